#!/bin/bash
# round 4: catch-up ahead -- parity tests of the one-call step, then A/B of C3-shaped / proxy / C4-shaped with and without it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lazy or one_call or g_step" 2>&1 | tail -5
for rep in 1 2; do
for ah in 1 pp0 0; do
  for wl in ml20m mid c4; do
    case $wl in
      ml20m) args="--workload ml20m --users 6400";;
      mid)   args="--workload custom:25024 --parallelism item-shard";;
      c4)    args="--workload c4 --users 3200";;
    esac
    a=$ah; pp=1; [ "$ah" = "pp0" ] && a=1 && pp=0; [ "$ah" = "0" ] && pp=0
    LTGAN_SHADOW_PINGPONG=$pp LTGAN_Q0_AHEAD=$a timeout 600 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r4/ahead_tmp.json 2> gpurun_out/r4/ahead_tmp.err || tail -3 gpurun_out/r4/ahead_tmp.err
    python - "$wl" "$ah" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4/ahead_tmp.json").read().strip().splitlines()[-1])
    n = d["config"]["batches"] * d["config"]["sub_epochs"]
    print("AB %-6s ahead=%-4s  users/s %7.0f  g_step_us %6.1f  d_step_us %5.1f" % (sys.argv[1], sys.argv[2], d["value"], d["phases_ms"]["t_g"] * 1e3 / n, d["phases_ms"]["t_d"] * 1e3 / n))
except Exception as e:
    print("AB", sys.argv[1], sys.argv[2], "failed", e)
PY
  done
done
done
