# round 6: the whole GPU suite on the current tree + smoke + one default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_full
mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['phases_ms'], d['config'].get('d_arith')); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','frac_source','avg_us_rocprof','avg_us_event_bracket')}); print(r.get('dominant_by_total_time')); ow=d['other_workloads']; print({k:(round(v.get('value',0)), round(v.get('g_step_us',0),1), round(v.get('d_step_us',0),1)) for k,v in ow.items()}); print(ow['rank_proxy'].get('exchanges_us'), ow['rank_proxy'].get('error')); print(d.get('projected_strong_scaling_8gpu_upper_bound'))"
