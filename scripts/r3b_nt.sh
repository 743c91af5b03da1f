# non-temporal theta / m / v accesses: the streaming decoder weight update (shipped) against plain accesses (-DLTG_DW_TEMPORAL) and against
# non-temporal m / v rows in the lazy clock's kernels as well (-DLTG_Q0_NT)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or g_step_parity" 2>&1 | tail -2
LTG_HIP_LIB=$R/build_ab/libltg_q0nt.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam" 2>&1 | tail -2
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
  run mid_temporal LTG_HIP_LIB=$R/build_ab/libltg_temporal.so -- $MID
  run mid_nt X=1 -- $MID
  run mid_nt_widegrad LTGAN_PIPE_FLAGS=8 -- $MID
  run mid_nt_q0nt LTG_HIP_LIB=$R/build_ab/libltg_q0nt.so -- $MID
done 2>&1 | grep "^AB" | tee $O/ab_nt_mid.txt
for rep in 1 2; do
  run c4_temporal LTG_HIP_LIB=$R/build_ab/libltg_temporal.so -- --workload c4 --users 3200
  run c4_nt X=1 -- --workload c4 --users 3200
  run c4_nt_q0nt LTG_HIP_LIB=$R/build_ab/libltg_q0nt.so -- --workload c4 --users 3200
  run c3_temporal LTG_HIP_LIB=$R/build_ab/libltg_temporal.so -- --workload ml20m --users 6400
  run c3_nt X=1 -- --workload ml20m --users 6400
  run c3_nt_q0nt LTG_HIP_LIB=$R/build_ab/libltg_q0nt.so -- --workload ml20m --users 6400
done 2>&1 | grep "^AB" | tee $O/ab_nt_other.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_sh -- python3 $R/bench.py $MID --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_sh.log 2>&1
cd $R
f=$(find $O/trace_sh -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_slice 2 > $O/timeline_nt.txt; rm -rf $O/trace_sh
cat $O/timeline_nt.txt
