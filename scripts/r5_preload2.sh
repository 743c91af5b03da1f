# round 5, experiment 16b: kernarg preload in the shipped build + PairView flattened into leading arguments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_preload2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "d_step or fp8 or precision" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= preload=$GRAFT_REPO_ROOT/ab_live/libltg_preload.so nopre=$GRAFT_REPO_ROOT/ab_live/libltg_prevnew.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
} 2>&1 | tee $O/ab.txt
