# repeat the tests that would expose a cross-stream hazard of the one-call sharded step (bit-identity against the dense sweep, 2-rank runs)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or (g_step_parity and one)" 2>&1 | tail -1
done
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -k "ml20m or c4" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -1
done
