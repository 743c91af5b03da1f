# round 4: ltg_d_step with jobs B / C of its backward on an aux stream (ltg_d_opts.aux_stream, Engine.d_fork) -- parity, same-box A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py tests/test_gpu_session.py tests/test_gpu_sharded.py -m gpu -q --timeout 1200 -k "d_step or trajectory_matches or session or expired or two_rank" 2>&1 | tail -8
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name env -- args
  name=$1; ev=$2; shift 2
  env $ev python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-22s users/s %7d  d_step_us %5.1f  g_step_us %6.1f  phases %s" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_d"] * 1e3 / (nb * 10), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), {k: round(v, 1) for k, v in d["phases_ms"].items()}))
PY
}
for rep in 1 2 3; do
  run d_fork LTGAN_D_FORK=1
  run d_one_stream LTGAN_D_FORK=0
  run c3_d_fork LTGAN_D_FORK=1 --workload ml20m --users 6400
  run c3_d_one_stream LTGAN_D_FORK=0 --workload ml20m --users 6400
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_d_fork.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_dk -- python3 $R/bench.py $B --steps 1 --warmup 1 > $O/tr_dk.log 2>&1
cd $R
f=$(find $O/tr_dk -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_d_l1 3 > $O/timeline_d_step.txt; rm -rf $O/tr_dk
head -40 $O/timeline_d_step.txt
