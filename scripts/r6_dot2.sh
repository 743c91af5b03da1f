# round 6: the split's residuals as v_dot2c_f32_bf16 (one instruction per element) -- exactness on both builds, then same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_dot2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16_split_is_exact or d_step_parity or forward_only_tower" > $O/pytest_new.log 2>&1; echo "new rc=$?"; tail -2 $O/pytest_new.log
LTG_HIP_LIB=$GRAFT_REPO_ROOT/ab_live/libltg_dot2.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16_split_is_exact or d_step_parity or forward_only_tower" > $O/pytest_dot2.log 2>&1; echo "dot2 rc=$?"; tail -2 $O/pytest_dot2.log
L="new= dot2=$GRAFT_REPO_ROOT/ab_live/libltg_dot2.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
} 2>&1 | tee $O/ab.txt
