cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "forward_only_tower" 2>&1 | grep -E "forward-only|passed|failed|Error" | tail -20
