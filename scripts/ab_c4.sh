# A/B on ONE box: bench a workload with the in-tree library (NEW) and build_ab/libltg_prev.so in the order P N P N N P
P=$GRAFT_REPO_ROOT/build_ab/libltg_prev.so
for lib in "$P" "" "$P" "" "" "$P"; do
    LTG_AB_COMPAT=1 LTG_HIP_LIB=$lib python bench.py --workload ${1:-c4} --users 3200 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
    python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); k=d['kernels_us']; print('%-8s' % ('${lib:-NEW}'[-7:]), round(d['value']), {n:k[n] for n in ('dec1_fwd','dh2','dec1_bwd_adam','enc0_bwd_adam')})"
done
