#!/bin/bash
# timeline of the one-call step with the catch-up ahead (per-rank proxy and the 20 000-item step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4
mkdir -p $R/$O
for wl in "mid25k:--workload custom:25024 --parallelism item-shard" "ml20m:--workload ml20m --users 3200"; do
  name=${wl%%:*}; extra=${wl#*:}
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$name -- python3 $R/bench.py $extra --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_$name.log 2>&1
  cd $R
  f=$(find $O/trace_$name -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_enc0_fwd 3 > $O/ahead_${name}_timeline.txt; rm -rf $O/trace_$name
  head -70 $O/ahead_${name}_timeline.txt
done
