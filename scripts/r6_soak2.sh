# round 6, final build: the soak's other slab sizes (ragged 65 544-item slab on the second streaming form, 9 000 items, a longer run at 25 024)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_soak
{
for a in "9000 20" "65544 10" "25024 40"; do echo "== items epochs: $a"; timeout 900 python scripts/soak_onecall.py $a 2>&1 | tail -3; done
} 2>&1 | tee gpurun_out/r6_soak/soak2.txt
