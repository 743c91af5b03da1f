# round 5: the second form of the streaming decoder forward -- parity, then same-box A/B against the first form (knob bit 26) + kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fwd
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "forward_parity or g_step_parity or streaming or lazy or warm_moments or trajectory or pipelined or hoisted or falls_back or expired" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2; do
for v in 0 1048576 67108864 68157440; do
  for wl in "c4:--workload c4 --users 3200" "ml20m:--workload ml20m --users 6400" "mid25k:--workload custom:25024 --parallelism item-shard"; do
    name=${wl%%:*}; extra=${wl#*:}
    python bench.py $extra --variant $v --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json
    python -c "
import json; d=json.load(open('$O/b.json')); print('variant %-9s %-7s' % ('$v', '$name'), round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, d.get('roofline', {}).get('kernel'), flush=True)" | tee -a $O/ab.txt
  done
done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 68157440; do
for wl in "c4:--workload c4 --users 1600" "ml20m:--workload ml20m --users 3200" "mid25k:--workload custom:25024 --parallelism item-shard --users 3200"; do
  name=${wl%%:*}; extra=${wl#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${name}_$v -- python3 $R/bench.py $extra --variant $v --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_${name}_$v.log 2>&1
  f=$(find $R/$O/prof_${name}_$v -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/${name}_${v}_kernel_stats.csv; rm -rf $R/$O/prof_${name}_$v
  grep -i "dec1_fwd_stream\|dh2_stream\|row_stats\|dlogits" $R/$O/${name}_${v}_kernel_stats.csv | cut -c1-200
done
done
