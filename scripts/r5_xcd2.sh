cd $GRAFT_REPO_ROOT
L="new= base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so tailbn64=$GRAFT_REPO_ROOT/ab_live/libltg_tailbn64.so"
echo "== askubuntu: base = before the fk_d_l1 order; tailbn64 = new + 64-column tail tiles"; bash scripts/ab_libs.sh "$L" --steps 10
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "d_step" 2>&1 | tail -2
