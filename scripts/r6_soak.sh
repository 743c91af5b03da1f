# round 6, final build: soak of the one-call G step (pipelined == step-by-step bit for bit, no expired wait) at three slab sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_soak
{
for a in "25024 20" "20000 15" "200000 3"; do echo "== items epochs: $a"; timeout 900 python scripts/soak_onecall.py $a 2>&1 | tail -3; done
} 2>&1 | tee gpurun_out/r6_soak/soak.txt
