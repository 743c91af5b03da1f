# round 5, experiment 11: dependent round trips taken out of the latency-path kernels (asm_trips.py audit)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_trips
mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu -k "not eight_rank and not two_ranks" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
L="new= base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
echo "== c4"; bash scripts/ab_libs.sh "$L" --workload c4 --users 3200
} 2>&1 | tee $O/ab.txt
