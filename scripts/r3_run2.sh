# round 3, GPU run 2: one-call sharded step after the shadow-race fix -- parity (incl. 2-rank gloo cases), weight-update workgroup sweep, stream priority
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "g_step_parity or lazy_adam" 2>&1 | tail -15 > $O/tests_parity.log
tail -3 $O/tests_parity.log
B="python bench.py --workload custom:25024 --parallelism item-shard --warm-moments --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe"
run() { # name, env...
  n=$1; shift
  env "$@" $B 2>$O/$n.err | tail -1 > $O/$n.json
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
run pipe_$rep X=1
run pipe_g160_$rep LTGAN_PIPE_FLAGS=$((160*256))
run pipe_g128_$rep LTGAN_PIPE_FLAGS=$((128*256))
run pipe_g112_$rep LTGAN_PIPE_FLAGS=$((112*256))
run pipe_g96_$rep LTGAN_PIPE_FLAGS=$((96*256))
run pipe_g224_$rep LTGAN_PIPE_FLAGS=$((224*256))
run pipe_prio_$rep LTGAN_G_PRIORITY=1
run pipe_prio_g128_$rep LTGAN_G_PRIORITY=1 LTGAN_PIPE_FLAGS=$((128*256))
done
timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -25 > $O/tests_sharded.log
tail -5 $O/tests_sharded.log
