set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run8
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp8_backward or forward_parity or g_step_parity or streaming or lazy" 2>&1 | tail -6 > $O/tests.log
tail -4 $O/tests.log
run() { # name, args...
  n=$1; shift
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe "$@" 2>$O/$n.err | tail -1 > $O/$n.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); nb=d["config"]["batches"]; S=d["config"]["sub_epochs"]
    print("$n", round(d["value"]), {k: round(v,2) for k,v in d["phases_ms"].items()}, "g_step_us %.1f d_step_us %.1f" % (d["phases_ms"]["t_g"]*1e3/(nb*S), d["phases_ms"]["t_d"]*1e3/(nb*S)))
except Exception as e:
    print("$n failed", e)
PY
}
for rep in 1 2; do
run c4_new_$rep --workload c4
run c4_old_$rep --workload c4 --variant 16384
run c4w_new_$rep --workload c4 --warm-moments
run c4w_old_$rep --workload c4 --warm-moments --variant 16384
run mid_new_$rep --workload custom:25024 --parallelism item-shard --warm-moments
run mid_old_$rep --workload custom:25024 --parallelism item-shard --warm-moments --variant 16384
run c3_new_$rep --workload ml20m
run c3_old_$rep --workload ml20m --variant 16384
done
cd /tmp
for v in 0 16384; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$v -- python3 $R/bench.py --workload c4 --users 3200 --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads --variant $v > $R/$O/prof$v.log 2>&1
f=$(find $R/$O/prof$v -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/c4_3200users_variant${v}_kernel_stats.csv; rm -rf $R/$O/prof$v
grep "dec1_fwd_stream\|dh2_stream\|dec1_bwd_adam_stream" $R/$O/c4_3200users_variant${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
done
