# round 5, experiment 12: streaming kernels' prologues (W tiles requested before / beside the h2 fragments, first tile stashed behind all requests),
# poison guards on the scalar unit
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_trips5
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or g_step or lazy or one_call or hoisted or span" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= c5be=$GRAFT_REPO_ROOT/ab_live/libltg_c5be.so"
{
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
echo "== c4"; bash scripts/ab_libs.sh "$L" --workload c4 --users 3200
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
} 2>&1 | tee $O/ab.txt
