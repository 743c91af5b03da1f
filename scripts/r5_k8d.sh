# round 5, experiment 18d: fk_d_l1 over two K slices (512 threads)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_k8d
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "d_step or g_step or fork or lazy or one_call" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= pre8d=$GRAFT_REPO_ROOT/ab_live/libltg_pre8d.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
