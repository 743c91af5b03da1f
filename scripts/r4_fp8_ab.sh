#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "fp8 or precision or wide" 2>&1 | tail -3
for rep in 1 2 3; do
for lib in tree old; do
  e=X=1; [ "$lib" != "tree" ] && e=LTG_HIP_LIB=$PWD/build_ab/libltg_three.so
  env $e timeout 600 python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/fp8_tmp.json
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4/fp8_tmp.json").read().strip().splitlines()[-1])
n = d["config"]["batches"] * d["config"]["sub_epochs"]
print("AB wide-fp8 %-5s users/s %7.0f  d_step_us %6.1f  g_step_us %5.1f  %s %.1f us" % (sys.argv[1], d["value"], d["phases_ms"]["t_d"] * 1e3 / n, d["phases_ms"]["t_g"] * 1e3 / n, d["roofline"]["kernel"], d["roofline"]["avg_us"]))
PY
done
done
