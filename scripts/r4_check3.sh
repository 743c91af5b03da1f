cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -6
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_default.json"))
print("DEFAULT askubuntu", round(d["value"]), {k: round(v, 1) for k, v in d["phases_ms"].items()})
for k, v in d["other_workloads"].items():
    print("DEFAULT", k, round(v["value"]), "g_step_us", round(v["g_step_us"], 1), "d_step_us", round(v["d_step_us"], 1), "step_frac", round(v["step_frac"], 3), v.get("handover"))
PY
done
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('STANDALONE c3', round(d['value']), d['phases_ms'])"
