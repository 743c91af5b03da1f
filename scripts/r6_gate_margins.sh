cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ndcg_gate.py -q -m gpu -s 2>&1 | grep -E "NDCG@100 after|curve, worst|wide discriminator, NDCG|worst relative|passed|failed"
