cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py -q -m gpu -k "one_shot" 2>&1 | tail -15
