# round 5: workgroups of the streaming weight update beside the (now shorter) chain, 20 000 items: 157 (the library's choice: 4 rounds of 625 tiles) against 209 (3 rounds) and others
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_dwg
for rep in 1 2 3; do for g in 0 209 180 167; do
  LTGAN_PIPE_FLAGS=$((g << 8)) python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads --no-probe --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/ab.json
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('groups %-4s' % '$g', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"
done; done 2>&1 | tee gpurun_out/r5_dwg/ab.txt
