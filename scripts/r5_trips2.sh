# round 5, experiment 11b: peel / no peel / base, same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_trips2
mkdir -p $O
L="new= nopeel=$GRAFT_REPO_ROOT/ab_live/libltg_nopeel.so base=$GRAFT_REPO_ROOT/ab_live/libltg_base.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
} 2>&1 | tee $O/ab.txt
