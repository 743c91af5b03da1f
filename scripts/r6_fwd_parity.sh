cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_forward_parity" 2>&1 | tail -3
