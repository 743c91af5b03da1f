# the D step's fork only from 1 024 pair rows: the three workloads again (new build; LTGAN_D_FORK=0 = never)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_dfork
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "d_step" 2>&1 | tail -1
run() { LTGAN_D_FORK=$2 python bench.py $3 --no-cpu-baseline --no-other-workloads --no-probe --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/ab.json
  python -c "
import json; d=json.load(open('$O/ab.json')); nb=d['config']['batches']*d['config']['sub_epochs']; print('%-10s %-34s' % ('$1','$3'), round(d['value']), 'D step us %.2f' % (d['phases_ms']['t_d']*1e3/nb), 'G step us %.2f' % (d['phases_ms']['t_g']*1e3/nb))"; }
{ for rep in 1 2 3; do run threshold 1 "--workload ml20m --users 6400"; run never 0 "--workload ml20m --users 6400"; done
  for rep in 1 2; do run threshold 1 "--workload c4 --users 6400"; run never 0 "--workload c4 --users 6400"; done
  for rep in 1 2 3; do run threshold 1 "--steps 10"; run never 0 "--steps 10"; done; } 2>&1 | tee $O/ab2.txt
