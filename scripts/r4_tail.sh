# round 4: the Adam tail on its own stream (ltg_pipe.tail_stream) -- parity, soak, same-box A/B against the tail on the caller's stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -q --timeout 1200 -k "expired or pipelined or falls_back or lazy_adam or one-call or one_call" 2>&1 | tail -6
for it in 25024 20000 9000; do timeout 600 python scripts/soak_onecall.py $it 3 2>&1 | tail -2; done
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
MID="--workload custom:25024 --parallelism item-shard"
C3="--workload ml20m --users 6400"
C4="--workload c4 --users 3200"
for rep in 1 2 3; do
  run mid_tail_own 0 $MID
  run mid_tail_inline 64 $MID
  run c3_tail_own 0 $C3
  run c3_tail_inline 64 $C3
  run c4_tail_own 0 $C4
  run c4_tail_inline 64 $C4
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_tail_stream.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_tail -- python3 $R/bench.py $B $MID --steps 1 --warmup 1 > $O/tr_tail.log 2>&1
cd $R
f=$(find $O/tr_tail -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_tail.txt; rm -rf $O/tr_tail
cat $O/timeline_tail.txt
