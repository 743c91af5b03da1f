# round 5, experiment 21: fk_d_bwd1 with 512 threads, eight K slices in jobs A and B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_bwd1k8
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_session.py -x -q -m gpu -k "trajectory or d_step or fork or hoisted or session or factories" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= prebwd1=$GRAFT_REPO_ROOT/ab_live/libltg_prebwd1.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
