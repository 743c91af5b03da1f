# round 4: theta / m / v of the streaming weight update as PLAIN accesses (the 180 MB of a 25 024-item slab fit the 256-MB memory-side cache:
# scripts/micro/adam_stream.hip reads 6.6 TB/s from 96 workgroups with plain accesses there, 3.9 with nt) against nt, by workgroup count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name lib flags -- args
  name=$1; lib=$2; fl=$3; shift 3
  LTG_HIP_LIB=$lib LTGAN_PIPE_FLAGS=$fl python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-26s users/s %7d  g_step_us %6.1f  d_step_us %5.1f" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), d["phases_ms"]["t_d"] * 1e3 / (nb * 10)))
PY
}
MID="--workload custom:25024 --parallelism item-shard --warm-moments"
C3="--workload ml20m --users 6400 --warm-moments"
for rep in 1 2; do
for g in 96 128 160 196; do
  run mid_nt_$g "" $((g << 8)) $MID
  run mid_plain_$g $R/build_ab/libltg_temporal.so $((g << 8)) $MID
  run c3_nt_$g "" $((g << 8)) $C3
  run c3_plain_$g $R/build_ab/libltg_temporal.so $((g << 8)) $C3
done
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_plain_vs_nt.txt
for v in nt plain; do
  lib=""; [ $v = plain ] && lib=$R/build_ab/libltg_temporal.so
  cd /tmp
  LTG_HIP_LIB=$lib LTGAN_PIPE_FLAGS=$((128 << 8)) rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$v -- python3 $R/bench.py $B $MID --steps 1 --warmup 1 > $O/tr_$v.log 2>&1
  cd $R
  f=$(find $O/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v, 128 workgroups, beside the chain"; cut -d, -f1-4 $f | sed 's/(anonymous namespace):://g' | cut -c1-150 | head -24
  f=$(find $O/tr_$v -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" k_q0_touch_unique 2 > $O/timeline_${v}_128.txt
  rm -rf $O/tr_$v
done
cat $O/timeline_plain_128.txt
