# kernel time under several knob values: bash scripts/r2_knobs.sh <kernel> "<v1> <v2> ..." [bench args]
cd $GRAFT_REPO_ROOT
KN=$1; VS=$2; shift; shift
for rep in 1 2; do for v in $VS; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads --variant $v "$@" 2>/dev/null | tail -1 > gpurun_out/knob.json
  python -c "
import json; d=json.load(open('gpurun_out/knob.json')); print('variant=%-9s' % '$v', round(d['value']), '$KN', d.get('kernels_us', {}).get('$KN'))"
done; done
