# final state of the round on one box: build entry, smoke, the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/tests_gpu.log
tail -2 $O/tests_gpu.log
python bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']), d['phases_ms'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['cpu_baseline']['value'], {k:round(v['value']) for k,v in d['other_workloads'].items()})"
