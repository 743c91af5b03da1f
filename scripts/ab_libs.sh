# same-box comparison of several builds of the library: bash scripts/ab_libs.sh "name=path name=path ..." [bench.py arguments...]
# (name "new" with an empty path = the in-tree library); three interleaved rounds
cd $GRAFT_REPO_ROOT
LIBS=$1; shift
for rep in 1 2 3; do
for kv in $LIBS; do
    name=${kv%%=*}; lib=${kv#*=}
    LTG_AB_COMPAT=1 LTG_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > gpurun_out/ab.json
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('%-8s' % '$name', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"
done
done
