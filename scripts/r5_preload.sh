# round 5, experiment 16: kernel arguments preloaded into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_preload
mkdir -p $O
LTG_HIP_LIB=$GRAFT_REPO_ROOT/ab_live/libltg_preload.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "d_step_parity or g_step_parity or forward_parity" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= preload=$GRAFT_REPO_ROOT/ab_live/libltg_preload.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
