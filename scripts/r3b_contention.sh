# What slows the chain beside the streaming weight update?  Two-step kernel timelines of the per-rank proxy with (a) the shipped library,
# (b) the update replaced by a dummy with its footprint and no memory traffic (LTG_X_SPIN), (c) by nothing
# (LTG_X_NOUPDATE).  MEASUREMENT BUILDS (scripts/build_variant.sh xspin -DLTG_X_SPIN=90; ... xnoupdate -DLTG_X_NOUPDATE): their results are
# wrong by design.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
MID="--workload custom:25024 --parallelism item-shard --warm-moments --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads"
for v in shipped xspin xnoupdate; do
  unset LTG_HIP_LIB LTGAN_PIPE_FLAGS
  [ $v = xspin ] && export LTG_HIP_LIB=$R/build_ab/libltg_xspin.so
  [ $v = xnoupdate ] && export LTG_HIP_LIB=$R/build_ab/libltg_xnoupdate.so
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- python3 $R/bench.py $MID > $O/trace_$v.log 2>&1
  cd $R
  f=$(find $O/trace_$v -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_slice 2 > $O/contention_$v.txt; rm -rf $O/trace_$v
  echo "== $v"; cat $O/contention_$v.txt
done
