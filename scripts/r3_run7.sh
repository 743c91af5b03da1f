set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp8 or precision_modes or d_step" 2>&1 | tail -15 > $O/tests_fp8.log
tail -12 $O/tests_fp8.log
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -k "wide" 2>&1 | tail -5
B="python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe"
run() { # name, env...
  n=$1; shift
  env "$@" $B $EXTRA 2>$O/$n.err | tail -1 > $O/$n.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); nb=d["config"]["batches"]; S=d["config"]["sub_epochs"]
    print("$n", round(d["value"]), {k: round(v,2) for k,v in d["phases_ms"].items()}, "g_step_us %.1f d_step_us %.1f" % (d["phases_ms"]["t_g"]*1e3/(nb*S), d["phases_ms"]["t_d"]*1e3/(nb*S)))
except Exception as e:
    print("$n failed", e)
PY
}
for rep in 1 2; do
EXTRA="" run wide_new_$rep X=1
EXTRA="--variant 524288" run wide_otf_$rep X=1
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/r3_askubuntu_wide_fp8_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3_run7/r3_askubuntu_wide_fp8_kernel_stats.csv')))
for r in rows[:22]:
    print("%-60s calls %5s avg %8.1f us tot %7.1f ms" % (r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
