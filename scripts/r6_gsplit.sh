# round 6: the six-term split in the generator's fp32 products that feed several MFMAs per fragment: the Adam tail's fp32 tiles (2 x 2 per wave) and enc-1 (1 x 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gsplit
mkdir -p $O
L="new= tail6=$GRAFT_REPO_ROOT/ab_live/libltg_tail6.so enc16=$GRAFT_REPO_ROOT/ab_live/libltg_enc16.so both6=$GRAFT_REPO_ROOT/ab_live/libltg_both6.so"
{
echo "== askubuntu"; bash scripts/ab_libs.sh "$L" --steps 10
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
} 2>&1 | tee $O/ab.txt
