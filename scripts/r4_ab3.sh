#!/bin/bash
# round 4, after catch-up ahead + two shadow buffers: where is the bound now?  tail stream on/off, workgroups of the weight update
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
run() {  # name env... -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r4/ab3_tmp.json 2> gpurun_out/r4/ab3_tmp.err || tail -3 gpurun_out/r4/ab3_tmp.err
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4/ab3_tmp.json").read().strip().splitlines()[-1])
    n = d["config"]["batches"] * d["config"]["sub_epochs"]
    print("AB %-22s users/s %7.0f  g_step_us %6.1f  d_step_us %5.1f" % (sys.argv[1], d["value"], d["phases_ms"]["t_g"] * 1e3 / n, d["phases_ms"]["t_d"] * 1e3 / n))
except Exception as e:
    print("AB", sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
  run c3_default X=1 -- --workload ml20m --users 6400
  run c3_tail_inline LTGAN_TAIL_STREAM=0 -- --workload ml20m --users 6400
  run c3_groups176 LTGAN_PIPE_FLAGS=$((176 << 8)) -- --workload ml20m --users 6400
  run c3_groups196 LTGAN_PIPE_FLAGS=$((196 << 8)) -- --workload ml20m --users 6400
  run mid_default X=1 -- --workload custom:25024 --parallelism item-shard
  run mid_tail_own LTGAN_PIPE_FLAGS=128 -- --workload custom:25024 --parallelism item-shard
  run mid_groups176 LTGAN_PIPE_FLAGS=$((176 << 8)) -- --workload custom:25024 --parallelism item-shard
  run mid_groups196 LTGAN_PIPE_FLAGS=$((196 << 8)) -- --workload custom:25024 --parallelism item-shard
  run mid_nocomm X=1 -- --workload custom:25024 --users 6400
done
