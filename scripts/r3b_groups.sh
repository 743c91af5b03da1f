# persistent workgroups of the forked weight update (ltg_pipe.flags bits 8-16), re-swept after the row-wave gradient kernel and the nt accesses
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3b
mkdir -p $O
MID="--workload custom:25024 --parallelism item-shard --warm-moments"
for rep in 1 2 3; do
for g in 0 224 196 176 157 144 131; do
  LTGAN_PIPE_FLAGS=$((g * 256)) python bench.py --no-cpu-baseline --no-other-workloads $MID 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json $g <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
s = d.get("sharded_step", {})
print("AB groups %4s users/s %7d  g_step_us %s" % (sys.argv[2], round(d["value"]), round(s["g_step_us"], 1)))
PY
done
done 2>&1 | grep "^AB" | sort -s -k3,3n | tee $O/ab_groups.txt
