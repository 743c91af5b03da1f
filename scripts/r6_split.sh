# round 6, item 1: the discriminator's fp32 GEMMs as bf16 cross terms of split operands (ltg_config.d_arith) -- parity first, then same-box A/B per kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_split
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "d_step" -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "D step|passed|failed|rc=" $O/pytest.log | tail -40
run() {   # name, d_arith, set
    LTGAN_D_ARITH=$2 LTGAN_D_ARITH_SET=$3 python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/ab.json
    python -c "
import json; d=json.load(open('$O/ab.json')); print('%-14s' % '$1', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, d['config'].get('d_arith'))"
}
{
for rep in 1 2 3; do
    run fp32 fp32 0
    run x6_l2b1b2 bf16x6 0xE
    run x6_all bf16x6 0xF
    run x6_l2 bf16x6 0x2
    run x6_bwd1 bf16x6 0x4
    run x6_bwd2 bf16x6 0x8
    run x4_l2b1b2 bf16x4 0xE
    run x4_all bf16x4 0xF
done
} 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in fp32 bf16x6 bf16x4; do
    LTGAN_D_ARITH=$v LTGAN_D_ARITH_SET=0xF rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
    f=$(find $GRAFT_REPO_ROOT/$O/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$O/kernel_stats_$v.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof_$v
    echo "== $v"; grep -E "fk_d_|fks_d" $GRAFT_REPO_ROOT/$O/kernel_stats_$v.csv | cut -c1-160
done
