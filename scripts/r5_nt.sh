cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_nt
L="new= tailnt=$GRAFT_REPO_ROOT/ab_live/libltg_tailnt.so"
{ echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10; } 2>&1 | tee gpurun_out/r5_nt/ab.txt
