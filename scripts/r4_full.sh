#!/bin/bash
# full GPU suite + smoke + the default bench line + the large-slab legs on the committed tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
python bench.py 2>gpurun_out/r4/full_bench.err | tail -1 > gpurun_out/r4/full_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/full_bench.json"))
print("headline", round(d["value"]), "users/s", d["ms_per_step"], "ms/epoch; roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
for k, v in d.get("other_workloads", {}).items():
    print(k, round(v["value"]), "users/s  g", round(v["g_step_us"], 1), "d", round(v["d_step_us"], 1), v.get("handover"))
PY
python bench.py --workload custom:25024 --parallelism item-shard --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r4/full_mid.json
python -c "
import json; d=json.load(open('gpurun_out/r4/full_mid.json')); print('proxy', round(d['value']), d.get('sharded_step'))"
