cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16_split_is_exact" 2>&1 | tail -5
