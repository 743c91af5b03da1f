# round 5, experiment 20: the dropout draw's pre-mix value formed once per thread in fk_d_l1 / fk_d_l2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_rng
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py -x -q -m gpu -k "trajectory or d_step or fork or hoisted" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= prerng=$GRAFT_REPO_ROOT/ab_live/libltg_prerng.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
} 2>&1 | tee $O/ab.txt
