# round 5, experiment 13: k8_d_adam with batched requests (config 5's Adam sweep)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fp8d
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp8 or precision_modes or d_step" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= c5be=$GRAFT_REPO_ROOT/ab_live/libltg_c5be.so"
{
echo "== askubuntu, wide fp8 discriminator"; bash scripts/ab_libs.sh "$L" --d-sizes 2048,1024,512,256 --d-precision fp8
} 2>&1 | tee $O/ab.txt
