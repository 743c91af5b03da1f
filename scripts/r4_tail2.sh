cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -q --timeout 1200 -k "expired or pipelined or falls_back or lazy_adam or fp8 or precision_modes" 2>&1 | tail -6
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name lib flags -- args
  name=$1; lib=$2; fl=$3; shift 3
  LTG_HIP_LIB=$lib LTGAN_PIPE_FLAGS=$fl python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-26s users/s %7d  g_step_us %6.1f  d_step_us %5.1f  %s" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), d["phases_ms"]["t_d"] * 1e3 / (nb * 10), d.get("sharded_step", {}).get("handover", "")))
PY
}
MID="--workload custom:25024 --parallelism item-shard"
C3="--workload ml20m --users 6400"
W8="--d-sizes 2048,1024,512,256 --d-precision fp8"
for rep in 1 2 3; do
  run mid_tail_own "" 0 $MID
  run mid_tail_inline "" 64 $MID
  run c3_tail_own "" 0 $C3
  run c3_tail_inline "" 64 $C3
  run wide8_deep "" 0 $W8
  run wide8_shallow $R/build_ab/libltg_sg8shallow.so 0 $W8
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_tail_stream2.txt
for v in deep shallow; do
lib=""; [ $v = shallow ] && lib=$R/build_ab/libltg_sg8shallow.so
cd /tmp
LTG_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_w8$v -- python3 $R/bench.py $B $W8 --steps 1 --warmup 1 > $O/tr_w8$v.log 2>&1
cd $R
f=$(find $O/tr_w8$v -name "*kernel_stats.csv" | head -1)
echo "== wide fp8, $v"; python - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-60s %6s calls  %8.2f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cp $f $O/r4_wide_fp8_${v}_kernel_stats.csv
rm -rf $O/tr_w8$v
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_tail -- python3 $R/bench.py $B $MID --steps 1 --warmup 1 > $O/tr_tail.log 2>&1
cd $R
f=$(find $O/tr_tail -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_tail.txt; rm -rf $O/tr_tail
grep -E "fk_g_tail|grad_rows|touch|enc0_fwd" $O/timeline_tail.txt
