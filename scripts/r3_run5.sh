set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run5
mkdir -p $O
python scripts/debug_onecall_bits.py 2>&1 | grep "^step" | cut -c1-600
LTGAN_TEST_PIPE_FLAGS=3 python scripts/debug_onecall_bits.py 2>&1 | grep "^step" | cut -c1-300
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
run pipe_g144_$rep LTGAN_PIPE_FLAGS=$((144*256))
run pipe_g176_$rep LTGAN_PIPE_FLAGS=$((176*256))
run pipe_g160_f2_$rep LTGAN_PIPE_FLAGS=$((160*256+2))
done
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -5
