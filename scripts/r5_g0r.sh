# round 5, experiment 22: fk_enc0_grad_rows with four instead of three entries in flight per wave (H = 600)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_g0r
mkdir -p $O
L="new= g0ru4=$GRAFT_REPO_ROOT/ab_live/libltg_g0ru4.so g0ru4b=$GRAFT_REPO_ROOT/ab_live/libltg_g0ru4b.so"
{
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
} 2>&1 | tee $O/ab.txt
