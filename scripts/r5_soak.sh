# round 5, final build: soak of the one-call G step (pipelined == step-by-step bit for bit, no expired wait) at four slab sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_soak
{
for a in "25024 40" "20000 30" "9000 30" "65544 15"; do echo "== items epochs: $a"; timeout 900 python scripts/soak_onecall.py $a 2>&1 | tail -3; done
echo "== items epochs: 200000 5"; timeout 900 python scripts/soak_onecall.py 200000 5 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r5_soak/soak.txt
