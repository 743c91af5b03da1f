#!/bin/bash
# one block for theta / m / v of W_p1t (LTGAN_ONE_BLOCK, default on) against three allocations, processes alternating
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
for i in 1 2 3 4 5 6 7 8; do
  ob=$((i % 2))
  LTGAN_ONE_BLOCK=$ob python bench.py --workload c4 --users 6400 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/mode2_$i.json
  python - $i $ob <<'PY'
import json, sys
i, ob = sys.argv[1:3]
d = json.load(open("gpurun_out/r4/mode2_%s.json" % i)); n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("run", i, "one_block =", ob, round(d["value"]), "users/s  G", round(d["phases_ms"]["t_g"] * 1e3 / n, 1), "us  update", round(d["roofline"]["avg_us"], 1), "us")
PY
done
for i in 1 2 3 4; do
  ob=$((i % 2))
  LTGAN_ONE_BLOCK=$ob python bench.py --workload custom:25024 --parallelism item-shard --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/mode2_m$i.json
  python - m$i $ob <<'PY'
import json, sys
i, ob = sys.argv[1:3]
d = json.load(open("gpurun_out/r4/mode2_%s.json" % i)); n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("mid run", i, "one_block =", ob, round(d["value"]), "users/s  G", round(d["phases_ms"]["t_g"] * 1e3 / n, 1), "us  update", round(d["roofline"]["avg_us"], 1), "us")
PY
done
