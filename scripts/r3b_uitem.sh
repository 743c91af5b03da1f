# Round 3, second half: distinct-item list in the batch (short dependent chains in the lazy clock kernels + sparse gradient), the clock
# slice in the catch-up launch.  Parity first, then same-box A/B on the per-rank proxy and the other workloads.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3b
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or g_step_parity or one_call" 2>&1 | tail -4 > $O/tests1.log
tail -2 $O/tests1.log
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sharded.py tests/test_gpu_session.py -m gpu -q -x 2>&1 | tail -4 > $O/tests2.log
tail -2 $O/tests2.log
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
s = d.get("sharded_step", {})
print("AB %-28s users/s %7d  g_step_us %s  phases %s" % (sys.argv[2], round(d["value"]), s.get("g_step_us") and round(s["g_step_us"], 1), {k: round(v, 1) for k, v in d.get("phases_ms", {}).items()}))
PY
}
MID="--workload custom:25024 --parallelism item-shard --warm-moments"
for rep in 1 2 3; do
  run mid_prev_lib LTG_HIP_LIB=$R/build_ab/libltg_prev.so -- $MID
  run mid_sideslice_nouitem LTGAN_PIPE_FLAGS=4 LTGAN_NO_UITEM=1 -- $MID
  run mid_merged_nouitem LTGAN_NO_UITEM=1 -- $MID
  run mid_merged_uitem X=1 -- $MID
  run mid_sideslice_uitem LTGAN_PIPE_FLAGS=4 -- $MID
  run mid_uitem_g0w1 LTG_HIP_LIB=$R/build_ab/libltg_g0w1.so -- $MID
  run mid_uitem_bn64 LTG_HIP_LIB=$R/build_ab/libltg_bn64.so -- $MID
done 2>&1 | grep "^AB" | tee $O/ab_mid.txt
for rep in 1 2; do
  run c3_nouitem LTGAN_NO_UITEM=1 -- --workload ml20m --users 6400
  run c3_uitem X=1 -- --workload ml20m --users 6400
  run c4_nouitem LTGAN_NO_UITEM=1 -- --workload c4 --users 3200
  run c4_uitem X=1 -- --workload c4 --users 3200
done 2>&1 | grep "^AB" | tee $O/ab_other.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_mid -- python3 $R/bench.py $MID --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_mid.log 2>&1
cd $R
f=$(find $O/trace_mid -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_slice 2 > $O/r3b_mid25k_timeline.txt; rm -rf $O/trace_mid
cat $O/r3b_mid25k_timeline.txt
