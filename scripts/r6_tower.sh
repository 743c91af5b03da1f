# round 6, item 2: the forward-only fake tower as ONE kernel (csrc/ltg_tower.h) -- parity, then same-box A/B against the three launches (tuning bit 20)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_tower
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "tower or hoisted or g_step_parity" -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "forward-only|passed|failed|rc=|Error|error" $O/pytest.log | tail -40
run() {   # name, variant
    python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 10 --warmup 2 --variant $2 2>/dev/null | tail -1 > $O/ab.json
    python -c "
import json; d=json.load(open('$O/ab.json')); print('%-14s' % '$1', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, d['config'].get('d_arith'))"
}
{
for rep in 1 2 3; do
    run fused 0
    run three_launch 1048576
done
} 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$O/kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof
grep -E "fkt_|fks_d|fk_d_y" $GRAFT_REPO_ROOT/$O/kernel_stats.csv | cut -c1-60,200-300
