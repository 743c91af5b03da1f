#!/bin/bash
cd "$GRAFT_REPO_ROOT"
V=${1:-three}; mkdir -p gpurun_out/r4
for rep in 1 2 3 4; do
for lib in tree $V; do
  e=X=1; [ "$lib" != "tree" ] && e=LTG_HIP_LIB=$PWD/build_ab/libltg_$lib.so
  env $e timeout 600 python bench.py --workload c4 --users 3200 --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/ab7_tmp.json
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4/ab7_tmp.json").read().strip().splitlines()[-1])
n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("AB c4 %-8s users/s %7.0f  g_step_us %6.1f  update %.1f us" % (sys.argv[1], d["value"], d["phases_ms"]["t_g"] * 1e3 / n, d["roofline"]["avg_us"]))
PY
done
done
