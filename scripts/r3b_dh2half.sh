# dh2 product split over column halves (half the partial slabs): parity, then same-box A/B against the library of the commit before
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3b
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or g_step_parity or one_call" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -3
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
  run mid_prev LTG_HIP_LIB=$R/build_ab/libltg_prev.so -- $MID
  run mid_halves X=1 -- $MID
  run c3_prev LTG_HIP_LIB=$R/build_ab/libltg_prev.so -- --workload ml20m --users 6400
  run c3_halves X=1 -- --workload ml20m --users 6400
  run c4_prev LTG_HIP_LIB=$R/build_ab/libltg_prev.so -- --workload c4 --users 3200
  run c4_halves X=1 -- --workload c4 --users 3200
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_dh2half.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_h -- python3 $R/bench.py $MID --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_h.log 2>&1
cd $R
f=$(find $O/prof_h -name "*kernel_stats.csv" | head -1); grep -i "dh2_stream\|k_da2" "$f" | cut -c1-200; rm -rf $O/prof_h
