# same library, two settings of the tuning knob (cfg.tuning) on ONE box: A/B of a schedule or kernel variant
# usage: ab_variant.sh <workload> <variantA> <variantB> [kernel names...]
W=${1:-c4}; A=${2:-0}; B=${3:-512}; shift 3 2>/dev/null
KS=${@:-dec1_bwd_adam enc0_bwd_adam dz dh1}
for v in $A $B $A $B; do
    python bench.py --workload $W --users 3200 --steps 2 --warmup 1 --no-cpu-baseline --variant $v 2>/dev/null | tail -1 > gpurun_out/ab.json
    python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); k=d['kernels_us']; print('variant %5d' % $v, round(d['value']), round(d['phases_ms']['t_g'],1), {n:k[n] for n in '$KS'.split()})"
done
