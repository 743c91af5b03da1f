# wide discriminator (BASELINE config 5 sizes) in fp8: operand-format storage (default) vs on-the-fly conversion (knob bit 18), same box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "precision_modes or fp8" 2>&1 | tail -3
for v in 0 262144 0 262144; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --d-sizes 2048,1024,512,256 --d-precision fp8 --variant $v 2>/dev/null | tail -1 > gpurun_out/fp8.json
  python -c "
import json; d=json.load(open('gpurun_out/fp8.json')); k=d.get('kernels_us',{}); print('variant', $v, round(d['value']), {x: round(y,1) for x,y in d['phases_ms'].items()}, {n: k.get(n) for n in ('d_l1','d_l2','d_bwd1','d_bwd2','d_adam')})"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fp8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads --d-sizes 2048,1024,512,256 --d-precision fp8 > $GRAFT_REPO_ROOT/gpurun_out/prof_fp8.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_fp8 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_fp8wide_kernel_stats.csv; rm -rf gpurun_out/prof_fp8
python3 -c "import csv; [print(r[\"Name\"][:50], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1)) for r in list(csv.DictReader(open(\"gpurun_out/r2_fp8wide_kernel_stats.csv\")))[:10]]"
