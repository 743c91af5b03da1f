# round 5, experiment 11d: epilogue operands of the generator's middle layers requested in the mid hook (behind the product's operands)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_trips4
mkdir -p $O
L="new= c5be=$GRAFT_REPO_ROOT/ab_live/libltg_c5be.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
} 2>&1 | tee $O/ab.txt
