# device words + the clock slice on the side stream behind gate kernels (the default) against the slice in the catch-up launch (LTG_PIPE_SLICE_IN_TOUCH = 32)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3b
mkdir -p $O
export LTGAN_PIPE_FLAGS=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or g_step_parity or one_call" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sharded.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -3
bash scripts/r3_stress.sh 2>&1 | grep -v "^+"
unset LTGAN_PIPE_FLAGS
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
  run mid_slice_in_touch LTGAN_PIPE_FLAGS=32 -- $MID
  run mid_slice_gated X=1 -- $MID
  run c3_slice_in_touch LTGAN_PIPE_FLAGS=32 -- --workload ml20m --users 6400
  run c3_slice_gated X=1 -- --workload ml20m --users 6400
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_gated_slice.txt
export LTGAN_PIPE_FLAGS=0
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_g -- python3 $R/bench.py $MID --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_g.log 2>&1
cd $R
f=$(find $O/trace_g -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_gated_slice.txt; rm -rf $O/trace_g
cat $O/timeline_gated_slice.txt
