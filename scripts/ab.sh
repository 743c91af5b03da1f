# A/B on ONE box: bench with the in-tree library (NEW) and build_ab/libltg_prev.so (PREV), alternating P N P N N P.
# usage: bash scripts/ab.sh [bench.py arguments...]     (default workload: Askubuntu_Sample)
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/build_ab/libltg_prev.so
for lib in "$P" "" "$P" "" "" "$P"; do
    LTG_AB_COMPAT=1 LTG_HIP_LIB=$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe "$@" 2>/dev/null | tail -1 > gpurun_out/ab.json
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('%-5s' % ('PREV' if '$lib' else 'NEW'), round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"
done
