#!/bin/bash
# LDS / memory-instruction counters of the Askubuntu_Sample step's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4_askpmc
mkdir -p $R/$O
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  cd /tmp
  rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/pmc_$tag.log 2>&1
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
    if not (k.startswith("fk_") or k.startswith("k_")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if max(n[(k, c)] for c in agg[k]) < 50: continue
    print("%-22s" % k[:22], {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
  rm -rf $O/pmc_$tag
done
