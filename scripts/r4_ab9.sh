#!/bin/bash
cd "$GRAFT_REPO_ROOT"
V=${1:-prev}
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "g_step or forward_parity or trajectory or pipelined or lazy" 2>&1 | tail -3
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r4/ab9_tmp.json 2> gpurun_out/r4/ab9_tmp.err || tail -3 gpurun_out/r4/ab9_tmp.err
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4/ab9_tmp.json").read().strip().splitlines()[-1])
    n = d["config"]["batches"] * d["config"]["sub_epochs"]
    print("AB %-12s users/s %7.0f  g_step_us %6.1f  d_step_us %5.1f  update %.1f us" % (sys.argv[1], d["value"], d["phases_ms"]["t_g"] * 1e3 / n, d["phases_ms"]["t_d"] * 1e3 / n, d["roofline"]["avg_us"]))
except Exception as e:
    print("AB", sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  for wl in "c3:--workload ml20m --users 6400" "mid:--workload custom:25024 --parallelism item-shard" "c4:--workload c4 --users 3200"; do
    name=${wl%%:*}; extra=${wl#*:}
    run ${name}_tree X=1 -- $extra
    run ${name}_$V LTG_HIP_LIB=$PWD/build_ab/libltg_$V.so -- $extra
  done
done
