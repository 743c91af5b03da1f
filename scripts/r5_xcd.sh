# round 5: XCD-aware tile order in the latency-path kernels.  tail3 = fk_g_tail + fk_dec1 + fk_dh2 only; new = also enc-1, dec-0, dz, dh1, D backward stage 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_xcd
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "g_step_parity or d_step or forward_parity or trajectory or pipelined or hoisted or forked or lazy" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= tail3=$GRAFT_REPO_ROOT/ab_live/libltg_tail3.so"
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
