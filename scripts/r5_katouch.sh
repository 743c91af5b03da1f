# round 5, experiment 17: every line of the kernel-argument segment requested at once at the top of the latency kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_katouch
mkdir -p $O
L="new= notouch=$GRAFT_REPO_ROOT/ab_live/libltg_notouch.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
