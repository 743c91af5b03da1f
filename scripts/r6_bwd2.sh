# round 6, review item 7: fk_d_bwd2 with a workgroup owning 4 / all 7 of the 16-row tiles of its (K slab, 32-column tile) -- parity, time, PMC traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_bwd2
mkdir -p $O
for t in 128 256; do LTGAN_TUNING=$t timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "d_step" > $O/pytest_$t.log 2>&1; echo "tuning $t rc=$?"; tail -1 $O/pytest_$t.log; done
run() { LTGAN_TUNING=$2 python bench.py --no-cpu-baseline --no-other-workloads --no-probe --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/ab.json
  python -c "
import json; d=json.load(open('$O/ab.json')); print('%-10s' % '$1', round(d['value']), {k: round(v, 2) for k, v in d['phases_ms'].items()})"; }
{ for rep in 1 2 3; do run tmw1 0; run tmw4 128; run tmw7 256; done; } 2>&1 | tee $O/ab.txt
for t in 0 128 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    LTGAN_TUNING=$t rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_${t}_$c -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/pmc_${t}_$c.log 2>&1
    cd $R
    f=$(find $O/pmc_${t}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $t $c <<'PY' | tee -a $O/pmc.txt
import csv, sys
tot=n=0
for r in csv.DictReader(open(sys.argv[1])):
    if "fk_d_bwd2" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[3]: tot+=float(r["Counter_Value"]); n+=1
print("tuning", sys.argv[2], sys.argv[3], "avg per launch (KB)", round(tot/max(n,1),1), "launches", n)
PY
    rm -rf $O/pmc_${t}_$c
  done
  cd /tmp
  LTGAN_TUNING=$t rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$t -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_$t.log 2>&1
  cd $R
  f=$(find $O/prof_$t -name "*kernel_stats.csv" | head -1); grep "fk_d_bwd2" "$f" | cut -c1-40,150-260 | tee -a $O/pmc.txt; rm -rf $O/prof_$t
done
