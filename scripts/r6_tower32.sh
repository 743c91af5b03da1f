# round 6: the one-kernel tower as 256-thread workgroups over 32 pair rows, two per CU (tuning bit 7) -- parity, same-box A/B, kernel average
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_tower32
mkdir -p $O
LTGAN_TUNING=128 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -q -m gpu -s -k "forward_only_tower or hoisted or g_step_parity" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "forward-only.*bf16x6|passed|failed" $O/pytest.log | tail -8
run() { LTGAN_TUNING=$2 python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/ab.json
  python -c "
import json; d=json.load(open('$O/ab.json')); print('%-10s' % '$1', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"; }
{ for rep in 1 2 3; do run rows64 0; run rows32 128; done; } 2>&1 | tee $O/ab.txt
for t in 0 128; do
  cd /tmp
  LTGAN_TUNING=$t rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$t -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_$t.log 2>&1
  cd $R
  f=$(find $O/prof_$t -name "*kernel_stats.csv" | head -1); grep "fkt_d_tower" "$f" | cut -c1-50,200-330 | tee -a $O/ab.txt; rm -rf $O/prof_$t
done
