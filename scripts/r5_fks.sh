# round 5, experiment 19: the dropout decision's row part resolved once per row in the forward-only tower kernels (fks_d_l1 / fks_d_l2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fks
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py -x -q -m gpu -k "hoisted or trajectory or d_step or g_step or tower or session" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= prefks=$GRAFT_REPO_ROOT/ab_live/libltg_prefks.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
