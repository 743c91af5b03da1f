# round 5, config 5: tile of the branch layers' product (knob bits 4-5: 16 = 64 x 64, 32 = 128 x 64, 48 = 128 x 128)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fp8b
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ndcg_gate.py -x -q -m gpu -k "fp8 or precision_modes" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -4
for rep in 1 2; do
for v in 16 32 48; do
  python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d=json.load(open('$O/b.json')); print('wide fp8 variant $v', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()}, flush=True)"
done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 16 32 48; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --variant $v --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof.log 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/wide_fp8_${v}_kernel_stats.csv; rm -rf $R/$O/prof
grep "fk8t_d_l1" $R/$O/wide_fp8_${v}_kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-60; grep "fk8t_d_l1" $R/$O/wide_fp8_${v}_kernel_stats.csv | awk -F'",' '{print $2}'
done
