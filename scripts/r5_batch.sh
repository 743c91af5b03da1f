# round 5: runtime-bound load loops (KL pairs in dec-0, y / fake-pair lists, gradient slabs, dh2 slabs) as clamped, masked batches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_batch
mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu -k "not eight_rank and not two_ranks" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
L="new= base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so"
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
echo "== c4"; bash scripts/ab_libs.sh "$L" --workload c4 --users 3200
