# round-2 GPU check: parity tests, then Askubuntu bench with the round-2 latency kernels and with the round-1 ones
# (ltg_config.tuning bit 18), then a kernel trace of each.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 1500 -x 2>&1 | tail -25 > gpurun_out/r2_tests.log
tail -5 gpurun_out/r2_tests.log
python bench.py --no-cpu-baseline --no-other-workloads --steps 5 --warmup 2 2>gpurun_out/r2_bench_new.err | tail -1 > gpurun_out/r2_bench_new.json
python bench.py --no-cpu-baseline --no-other-workloads --steps 5 --warmup 2 --variant 262144 2>gpurun_out/r2_bench_old.err | tail -1 > gpurun_out/r2_bench_old.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_new -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/gpurun_out/prof_new.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_old -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads --variant 262144 > $R/gpurun_out/prof_old.log 2>&1
cd $R
for d in prof_new prof_old; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_${d}_kernel_stats.csv; rm -rf gpurun_out/$d; done
python - <<'PY'
import json
for n in ("new", "old"):
    try:
        d = json.load(open("gpurun_out/r2_bench_%s.json" % n))
        print(n, round(d["value"]), d["phases_ms"], d.get("kernels_us"))
    except Exception as e:
        print(n, "failed", e)
PY
