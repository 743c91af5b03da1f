# round 4: GPU suite + the per-rank proxy / C3-shaped step by workgroup count of the weight update with the round's hand-over
# (h2 free behind the update's prologue, one-thread tail waits)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 -x --durations=8 2>&1 | tail -25
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
for rep in 1 2; do
for g in 0 112 128 144 160 176 196; do
  run mid_$g $((g << 8)) $MID
  run c3_$g $((g << 8)) $C3
done
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_groups_new_handover.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_new -- python3 $R/bench.py $B $MID --steps 1 --warmup 1 > $O/tr_new.log 2>&1
cd $R
f=$(find $O/tr_new -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_new.txt; rm -rf $O/tr_new
cat $O/timeline_new.txt
