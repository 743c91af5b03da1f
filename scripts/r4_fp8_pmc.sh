#!/bin/bash
# counters of the wide fp8 discriminator's branch-layer kernel (fk8t_d_l1): what is it waiting for?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4_fp8
mkdir -p $R/$O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  cd /tmp
  rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$tag -- python3 $R/bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/pmc_$tag.log 2>&1
  cd $R
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not k.startswith("fk8") and not k.startswith("k8") : continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k, {c: round(v / n[(k, c)], 1) for c, v in agg[k].items()}, "launches", max(n[(k, c)] for c in agg[k]))
PY
  rm -rf $O/pmc_$tag
done
