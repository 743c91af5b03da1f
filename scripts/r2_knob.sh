# same-box A/B of a tuning knob: bash scripts/r2_knob.sh <knob value> <kernel name in kernels_us> [bench args...]
cd $GRAFT_REPO_ROOT
K=$1; KN=$2; shift; shift
for v in $K 0 $K 0 $K 0; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --variant $v "$@" 2>/dev/null | tail -1 > gpurun_out/knob.json
  python -c "
import json; d=json.load(open('gpurun_out/knob.json')); print('variant=%-9s' % '$v', round(d['value']), {k: round(v, 1) for k, v in d['phases_ms'].items()}, '$KN', d.get('kernels_us', {}).get('$KN'))"
done
