# round 4, first measurements: (1) what an Adam sweep reaches from N workgroups (scripts/micro/adam_stream.hip); (2) today's box: the
# per-rank proxy, C3- and C4-shaped steps; (3) the streaming weight update ALONE (everything in program order) by workgroup count
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 300 ./build_ab/adam_stream 25024 2>&1 | tee $O/adam_stream_25024.txt
timeout 300 ./build_ab/adam_stream 200000 2>&1 | tee $O/adam_stream_200000.txt
MID="--workload custom:25024 --parallelism item-shard --warm-moments"
B="--no-cpu-baseline --no-other-workloads"
python bench.py $B $MID 2>/dev/null | tail -1 > $O/base_mid.json
python bench.py $B --workload ml20m --users 6400 --warm-moments 2>/dev/null | tail -1 > $O/base_c3.json
python bench.py $B --workload c4 --users 6400 --warm-moments 2>/dev/null | tail -1 > $O/base_c4.json
python - <<'PY'
import json
for n in ("mid", "c3", "c4"):
    d = json.load(open("gpurun_out/r4/base_%s.json" % n))
    s = d.get("sharded_step", {})
    print("BASE", n, round(d["value"]), "users/s; g_step_us", s.get("g_step_us"), "phases", {k: round(v, 1) for k, v in d["phases_ms"].items()}, "batches", d["config"]["batches"])
PY
for g in 98 128 160 196 224; do
  cd /tmp
  LTGAN_PIPE_FLAGS=$((1 + (g << 8))) rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone_$g -- python3 $R/bench.py $B $MID --steps 1 --warmup 1 --no-probe > $O/alone_$g.log 2>&1
  cd $R
  f=$(find $O/alone_$g -name "*kernel_stats.csv" | head -1)
  echo "ALONE groups=$g $(grep -h 'k_dec1_bwd_adam_stream' $f | head -1 | cut -c1-200)"
  rm -rf $O/alone_$g
done
