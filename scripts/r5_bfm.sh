# round 5: the three decoder GEMMs of small slabs on the bf16 matrix pipe (their operands were bf16-rounded all along)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_bfm
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py tests/test_gpu_ndcg_gate.py tests/test_gpu_session.py -x -q -m gpu -k "not fp8 and not 200000 and not 200008" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
L="new= base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so"
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
