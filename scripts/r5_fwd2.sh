cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fwd2
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "forward_parity or g_step_parity or streaming or lazy or warm_moments or pipelined or hoisted" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
LIBS="new=" VARIANTS="0 131072 67108864" bash scripts/r5_fwd_alone.sh 2>&1 | grep -v "^\[" | grep "==\|fwd_stream"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 67108864; do
  python bench.py --workload c4 --users 3200 --variant $v --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d=json.load(open('$O/b.json')); print('variant %-9s c4' % '$v', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, flush=True)"
done
done
