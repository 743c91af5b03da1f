cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_kernarg
for rep in 1 2; do
for v in unset 1 0; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/ab.json
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('HIP_FORCE_DEV_KERNARG=%-6s' % '$v', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"
done
done 2>&1 | tee gpurun_out/r5_kernarg/ab.txt
