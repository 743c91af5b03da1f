# round 5, experiment 11c: peel in the discriminator's kernels only + enc-0 gather with uniform branches; I-cache microbenchmark
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_trips3
mkdir -p $O
./ab_live/icache 2>&1 | tee $O/icache.txt
L="new= nopeel=$GRAFT_REPO_ROOT/ab_live/libltg_nopeel.so base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
