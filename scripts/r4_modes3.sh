#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
U=${1:-300000}
for i in 1 2 3 4 5 6; do
  ob=$((i % 2))
  LTGAN_ONE_BLOCK=$ob python bench.py --workload c4 --users $U --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/mode3_$i.json
  python - $i $ob <<'PY'
import json, sys
i, ob = sys.argv[1:3]
d = json.load(open("gpurun_out/r4/mode3_%s.json" % i)); n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("run", i, "users", d["config"]["users"], "one_block =", ob, round(d["value"]), "users/s  G", round(d["phases_ms"]["t_g"] * 1e3 / n, 1), "us  update", round(d["roofline"]["avg_us"], 1), "us")
PY
done
