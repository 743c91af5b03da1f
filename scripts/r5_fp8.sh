# round 5, config 5: the branch layers without the fp32 copy of A1 -- parity of every fp8 test, then D-step time + kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fp8
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ndcg_gate.py -x -q -m gpu -s -k "fp8 or precision_modes or gemm_block" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -12
for rep in 1 2; do
  python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d=json.load(open('$O/b.json')); print('wide fp8', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, flush=True)"
done
cp $O/b.json $O/r5_bench_askubuntu_wide_fp8.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof.log 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/r5_askubuntu_wide_fp8_kernel_stats.csv; rm -rf $R/$O/prof
head -12 $R/$O/r5_askubuntu_wide_fp8_kernel_stats.csv | cut -c1-60,200-300
