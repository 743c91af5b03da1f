# round 4: ltg_g_step on small slabs with two Adam jobs on the aux stream (ltg_g_opts.sync, Engine.g_tail_fork): parity, same-box A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_session.py tests/test_gpu_ndcg_gate.py -m gpu -q --timeout 1200 -k "small_slab or trajectory_matches or g_step_parity or session or injected or (ndcg and not full and not long)" 2>&1 | tail -8
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
  run g_tail_fork LTGAN_G_TAIL_FORK=1
  run g_one_launch LTGAN_G_TAIL_FORK=0
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_g_tail_fork.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_gk -- python3 $R/bench.py $B --steps 1 --warmup 1 > $O/tr_gk.log 2>&1
cd $R
f=$(find $O/tr_gk -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_enc0_fwd 3 > $O/timeline_g_step.txt; rm -rf $O/tr_gk
head -40 $O/timeline_g_step.txt
