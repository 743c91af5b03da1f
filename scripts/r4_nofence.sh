# round 4: the device words stored WITHOUT an agent-scope release (every producer is a whole kernel that ended before the store): soak
# (the pipelined loop against the step-by-step loop, bit for bit), GPU suite, same-box A/B against the build that fences (-DLTG_GATE_FENCE)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
for it in 25024 20000 9000 8200; do timeout 900 python scripts/soak_onecall.py $it 8 2>&1 | tail -2; done
timeout 900 python scripts/soak_onecall.py 200000 2 2>&1 | tail -2
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -5
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name lib -- args
  name=$1; lib=$2; shift 2
  LTG_HIP_LIB=$lib python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-22s users/s %7d  d_step_us %5.1f  g_step_us %6.1f" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_d"] * 1e3 / (nb * 10), d["phases_ms"]["t_g"] * 1e3 / (nb * 10)))
PY
}
MID="--workload custom:25024 --parallelism item-shard"
C3="--workload ml20m --users 6400"
for rep in 1 2 3; do
  run ask_nofence ""
  run ask_fence $R/build_ab/libltg_gatefence.so
  run mid_nofence "" $MID
  run mid_fence $R/build_ab/libltg_gatefence.so $MID
  run c3_nofence "" $C3
  run c3_fence $R/build_ab/libltg_gatefence.so $C3
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_gate_fence.txt
