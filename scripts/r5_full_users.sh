# the named configurations at their full user counts (C3: 136 000 users x 20 000 items; C4 on ONE GPU: 1 000 000 users x 200 000 items)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_full_users
mkdir -p $O
python bench.py --workload ml20m --users 136000 --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r5_bench_ml20m_full_136k_users.json
python bench.py --workload c4 --users 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r5_bench_c4_full_1m_users.json
python -c "
import json
for n in ('ml20m_full_136k_users','c4_full_1m_users'):
    d=json.load(open('$O/r5_bench_%s.json'%n)); print(n, round(d['value']), 'ms/step %.1f'%d['ms_per_step'], {k: round(v,1) for k,v in d['phases_ms'].items()})"
