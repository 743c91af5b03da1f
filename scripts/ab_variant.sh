# same library, two settings of the tuning knob (cfg.reserved0) on ONE box: A/B of a schedule
for v in ${2:-0} ${3:-512} ${2:-0} ${3:-512}; do
    python bench.py --workload ${1:-c4} --users 3200 --steps 2 --warmup 1 --no-cpu-baseline --variant $v 2>/dev/null | tail -1 > gpurun_out/ab.json
    python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); k=d['kernels_us']; print('variant %4d' % $v, round(d['value']), d['phases_ms']['t_g'], {n:k[n] for n in ('dec1_bwd_adam','enc0_bwd_adam','dz','dh1')})"
done
