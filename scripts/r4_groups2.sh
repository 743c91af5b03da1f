#!/bin/bash
# round 4: with the catch-up off the critical stream the cycle update -> streaming forward -> dlogits -> dh2 -> update bounds the step:
# persistent workgroups of the weight update again (bits 8-16 of the pipe's flags)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
for rep in 1 2; do
for g in 0 144 176 196 224; do
  for wl in ml20m mid; do
    case $wl in
      ml20m) args="--workload ml20m --users 6400";;
      mid)   args="--workload custom:25024 --parallelism item-shard";;
    esac
    LTGAN_PIPE_FLAGS=$((g << 8)) timeout 600 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r4/g2_tmp.json 2> gpurun_out/r4/g2_tmp.err || tail -3 gpurun_out/r4/g2_tmp.err
    python - "$wl" "$g" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4/g2_tmp.json").read().strip().splitlines()[-1])
    n = d["config"]["batches"] * d["config"]["sub_epochs"]
    print("AB %-6s groups=%-4s  users/s %7.0f  g_step_us %6.1f  d_step_us %5.1f" % (sys.argv[1], sys.argv[2], d["value"], d["phases_ms"]["t_g"] * 1e3 / n, d["phases_ms"]["t_d"] * 1e3 / n))
except Exception as e:
    print("AB", sys.argv[1], sys.argv[2], "failed", e)
PY
  done
done
done
