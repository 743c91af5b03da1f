cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=${c%% *}
  cd /tmp
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_fp8_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads --d-sizes 2048,1024,512,256 --d-precision fp8 > $R/gpurun_out/pmc_fp8_$tag.log 2>&1
  cd $R
  f=$(find gpurun_out/pmc_fp8_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:24]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
    print(k, {c: round(v/cnt[k][c],1) for c,v in agg[k].items()}, dict(cnt[k]))
PY
  rm -rf gpurun_out/pmc_fp8_$tag
done
