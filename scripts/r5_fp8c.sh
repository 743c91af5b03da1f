# round 5, config 5: three K blocks of global loads in flight in the staged e4m3 block (ltg_sgemm8_core, -DLTG_SG8_THREE)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fp8c
mkdir -p $O
LTG_HIP_LIB=$GRAFT_REPO_ROOT/ab_live/libltg_sg8three.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp8 or precision_modes" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -3
bash scripts/ab_libs.sh "new= three=$GRAFT_REPO_ROOT/ab_live/libltg_sg8three.so" --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 5
