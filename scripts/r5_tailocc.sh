# round 5: occupancy of the tail launch (fk_g_tail) -- __launch_bounds__(256, w): w = 5 (92 VGPRs, no scratch), 6 (80 + 40 B scratch), 7 (72 + 100 B); new = default bounds (96 + 16 AGPRs: 4 waves per SIMD)
cd $GRAFT_REPO_ROOT
L="new= tail5=$GRAFT_REPO_ROOT/ab_live/libltg_tail5.so tail6=$GRAFT_REPO_ROOT/ab_live/libltg_tail6.so tail7=$GRAFT_REPO_ROOT/ab_live/libltg_tail7.so"
bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "new= tail5=$GRAFT_REPO_ROOT/ab_live/libltg_tail5.so" --workload ml20m --users 6400
