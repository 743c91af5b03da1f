# round 5: forward-only towers (the batched fake tower of phase G) on the register-resident kernels (knob bit 23) against the LDS-staged ones
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_knob23
for rep in 1 2 3; do for v in 0 8388608; do
  python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 5 --warmup 1 --variant $v 2>/dev/null | tail -1 > gpurun_out/ab.json
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('variant %-8s' % '$v', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"
done; done 2>&1 | tee gpurun_out/r5_knob23/ab.txt
