set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_run11
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -m gpu -q -x -k "g_step_parity or trajectory or injected or lazy or slot_cache" 2>&1 | tail -3
bash scripts/ab.sh 2>&1 | tail -8
bash scripts/ab.sh --workload ml20m 2>&1 | tail -8
