# lazy Adam clock of W_q0 vs the dense sweep, same box: C4-shaped and ML-20M-shaped 1-GPU workloads
# (EXTRA="--warm-moments": every row of W_q0 has non-zero moments = the clock's full deferred arithmetic)
cd $GRAFT_REPO_ROOT
for wl in ${WLS:-c4 ml20m}; do
  for lazy in 0 1 0 1; do
    LTGAN_LAZY_Q0=$lazy python bench.py --workload $wl --users 6400 --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads $EXTRA 2>/dev/null | tail -1 > gpurun_out/lazy_${wl}_${lazy}.json
    python -c "
import json; d=json.load(open('gpurun_out/lazy_${wl}_${lazy}.json')); print('$wl lazy=$lazy', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, {k: v for k, v in d['kernels_us'].items() if 'enc0' in k or 'dec1_bwd' in k}, round(d['roofline']['step_frac'], 3), d['roofline'].get('lazy_q0'))"
  done
done
