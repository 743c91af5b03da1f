# round 6: the D step's aux-stream fork at the synthetic tables' ~300 pair rows per step (C3 / C4 / per-rank proxy): does it still pay there?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_dfork
mkdir -p $O
run() { LTGAN_D_FORK=$2 python bench.py --workload $3 --users 6400 --no-cpu-baseline --no-other-workloads --no-probe --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/ab.json
  python -c "
import json; d=json.load(open('$O/ab.json')); nb=d['config']['batches']*d['config']['sub_epochs']; print('%-8s %-6s' % ('$1','$3'), round(d['value']), 'D step us %.2f' % (d['phases_ms']['t_d']*1e3/nb), 'G step us %.2f' % (d['phases_ms']['t_g']*1e3/nb))"; }
{ for rep in 1 2 3; do run fork 1 ml20m; run nofork 0 ml20m; done; for rep in 1 2; do run fork 1 c4; run nofork 0 c4; done; } 2>&1 | tee $O/ab.txt
