set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run6
mkdir -p $O
python scripts/debug_onecall_bits.py 2>&1 | grep "^step" | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/tests_gpu.log
tail -4 $O/tests_gpu.log
python bench.py --workload custom:25024 --parallelism item-shard --warm-moments --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>$O/mid25k.err | tail -1 > $O/bench_mid25k.json
python -c "
import json; d=json.load(open('$O/bench_mid25k.json')); print(round(d['value']), d['phases_ms'], json.dumps(d.get('sharded_step'))[:1500]); print(d['roofline'])"
python bench.py --steps 5 --warmup 2 --cpu-seconds 5 2>$O/ask.err | tail -1 > $O/bench_ask.json
python -c "
import json; d=json.load(open('$O/bench_ask.json')); print(round(d['value']), d['phases_ms'], d['roofline']['kernel'], d['roofline']['frac'], {k:(round(v['value']),round(v['g_step_us'],1)) for k,v in d['other_workloads'].items()}, d.get('c4_same_workload'))"
