# round 4: slabs >= 65 536 items -- weight update and streaming forward as two launches each (word 11): parity, soak (bit-identity with the
# step-by-step loop), same-box A/B against one launch each (LTG_PIPE_NO_SPLIT = 131072)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -q --timeout 1200 -k "expired or 200000 or 200008 or c4 or falls_back" --durations=4 2>&1 | tail -12
for it in 200000 66016; do timeout 900 python scripts/soak_onecall.py $it 1 2>&1 | tail -2; done
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name flags -- args
  name=$1; fl=$2; shift 2
  LTGAN_PIPE_FLAGS=$fl python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-26s users/s %7d  g_step_us %6.1f  d_step_us %5.1f" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), d["phases_ms"]["t_d"] * 1e3 / (nb * 10)))
PY
}
C4="--workload c4 --users 3200"
for rep in 1 2 3; do
  run c4_two_launches 0 $C4
  run c4_one_launch 131072 $C4
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_split_c4.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_c4 -- python3 $R/bench.py $B --workload c4 --users 1600 --steps 1 --warmup 1 > $O/tr_c4.log 2>&1
cd $R
f=$(find $O/tr_c4 -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_c4.txt; rm -rf $O/tr_c4
cat $O/timeline_c4.txt
