#!/bin/bash
# the two modes of the 200 000-item step (weight update 571 vs 615-621 us, process after process on one box): clocks / power while it runs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
for i in 1 2 3 4 5 6; do
  python bench.py --workload c4 --users 6400 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/mode_$i.json &
  pid=$!
  sleep 14
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s ' ' | head -12 > gpurun_out/r4/mode_$i.smi
  wait $pid
  python - $i <<'PY'
import json, sys
i = sys.argv[1]
d = json.load(open("gpurun_out/r4/mode_%s.json" % i)); n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("run", i, round(d["value"]), "users/s  G", round(d["phases_ms"]["t_g"] * 1e3 / n, 1), "us  update", round(d["roofline"]["avg_us"], 1), "us")
print("   ", " | ".join(l.strip() for l in open("gpurun_out/r4/mode_%s.smi" % i).read().splitlines()))
PY
done
