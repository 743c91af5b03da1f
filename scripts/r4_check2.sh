cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 --durations=6 2>&1 | tail -14
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name flags -- args
  name=$1; fl=$2; shift 2
  LTGAN_PIPE_FLAGS=$fl python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-26s users/s %7d  g_step_us %6.1f  d_step_us %5.1f  %s" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), d["phases_ms"]["t_d"] * 1e3 / (nb * 10), d.get("sharded_step", {}).get("handover", "")))
PY
}
MID="--workload custom:25024 --parallelism item-shard"
C3="--workload ml20m --users 6400"
for rep in 1 2 3; do
  run askubuntu 0
  run mid_default 0 $MID
  run mid_tail_own 128 $MID
  run c3_default 0 $C3
  run c3_tail_inline 64 $C3
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/ab_tail_stream3.txt
