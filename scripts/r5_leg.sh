# why does the C3 leg of the default bench line run slower than the same workload alone?  (r5: 154 against 138 us per G step on one box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_leg
mkdir -p $O
show() { python -c "
import json; d=json.load(open('$O/b.json')); ow=d.get('other_workloads',{})
print('$1', round(d['value']), {k:(round(v['value']), round(v['g_step_us'],1), round(v['d_step_us'],1), v.get('handover')) for k,v in ow.items()}, {k: round(v, 2) for k, v in d['phases_ms'].items()}, flush=True)"; }
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json; show default
LTGAN_D_FORK=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json; show no_d_fork
GPU_MAX_HW_QUEUES=8 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json; show hwq8
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 > $O/b.json; show no_probe
python bench.py --workload ml20m --users 6400 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json; show ml20m_alone
LTGAN_D_FORK=0 python bench.py --workload ml20m --users 6400 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json; show ml20m_alone_no_d_fork
echo "== tail / dec1 / dh2 tile order (prevtail = the build before xcd_chunk in fk_g_tail, fk_dec1, fk_dh2)"
bash scripts/ab_libs.sh "new= prevtail=$GRAFT_REPO_ROOT/ab_live/libltg_prevtail.so" --steps 10
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "g_step_parity and 1000 or forward_parity and 1000 or g_step_parity and 333" 2>&1 | tail -2
