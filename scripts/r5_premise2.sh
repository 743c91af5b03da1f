# round 5, final build: the premise test of the tile-level hand-over AGAIN (the streaming forward not waiting for the previous update's end: results wrong, timing
# valid = the upper bound of any scheme that opens word 7 sooner) -- the caller's stream is ~10 us shorter than when it was first measured
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_premise2
mkdir -p $O
L="new= now7=$GRAFT_REPO_ROOT/ab_live/libltg_now7.so"
{
echo "== ml20m (20 000 items, no communicator)"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard (RCCL at world size 1)"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
} 2>&1 | tee $O/ab.txt
