# timing experiments through tuning-knob bits (ltg_config.tuning): kernel trace of one bench step per variant
# usage: EXTRA="--workload c4 --users 3200" bash scripts/r2_exp.sh <variant>...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-exp}
for v in "$@"; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v$v -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads --variant $v $EXTRA > $R/gpurun_out/prof_v$v.log 2>&1
  cd $R
  f=$(find gpurun_out/prof_v$v -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_${TAG}_v${v}_kernel_stats.csv; rm -rf gpurun_out/prof_v$v
  grep -h '^{' gpurun_out/prof_v$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant', $v, round(d['value']), d['phases_ms'])"
done
