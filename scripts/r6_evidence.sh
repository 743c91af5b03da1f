# Round-6 FINAL evidence, one box: bench JSON lines, kernel summaries (gate kernels kept out of the work percentages: profiles/clean_kernel_stats.py),
# timelines, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains) for four workloads, the D-step counter passes and the
# bf16-vs-pure-fp32-oracle probability gap.  Everything lands in gpurun_out/r6_evidence/ (scripts/r6_collect.sh copies the judged files into profiles/).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_evidence
mkdir -p $O
python bench.py --steps 20 --warmup 5 2>$O/bench_askubuntu.err | tail -1 > $O/r6_bench_askubuntu.json
python bench.py --workload c4 --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_ml20m.json
python bench.py --workload custom:25024 --parallelism item-shard --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_mid25k_item_shard.json
python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_askubuntu_wide_fp8.json
python bench.py --d-arith fp32 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_askubuntu_d_arith_fp32.json
for wl in "askubuntu:" "c4_3200users:--workload c4 --users 3200" "ml20m_3200users:--workload ml20m --users 3200" "mid25k:--workload custom:25024 --parallelism item-shard" "askubuntu_wide_fp8:--d-sizes 2048,1024,512,256 --d-precision fp8"; do
  name=${wl%%:*}; extra=${wl#*:}
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -- python3 $R/bench.py $extra --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_$name.log 2>&1
  cd $R
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); python profiles/clean_kernel_stats.py "$f" $O/r6_${name}_kernel_stats.csv; rm -rf $O/prof_$name
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_ask -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_ask.log 2>&1
cd $R
f=$(find $O/trace_ask -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_enc0_fwd 2 > $O/r6_askubuntu_g_step_timeline.txt; python profiles/make_timeline.py "$f" fk_d_l1 2 > $O/r6_d_step_timeline.txt; rm -rf $O/trace_ask
for wl in "askubuntu:" "c4:--workload c4 --users 1600" "ml20m:--workload ml20m --users 1600" "custom_25024:--workload custom:25024 --parallelism item-shard --users 1600"; do
  name=${wl%%:*}; extra=${wl#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp
    rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_${name}_$c -- python3 $R/bench.py $extra --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/pmc_${name}_$c.log 2>&1
    cd $R
    f=$(find $O/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$O/r6_${name}_$c.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
    rm -rf $O/pmc_${name}_$c
  done
done
# the discriminator step's kernels and the one-kernel tower: wave-cycle budget, MFMA busy time, cache hit rates (Askubuntu_Sample)
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  cd /tmp
  rocprofv3 --pmc $c --output-format csv -d $R/$O/dpmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/dpmc_$tag.log 2>&1
  cd $R
  f=$(find $O/dpmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O/r6_d_step_counters_raw.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    if not k.startswith(("fk_d_", "fkt_")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    print("%-22s launches %d" % (k[:22], max(n[(k, c)] for c in agg[k])), {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
  rm -rf $O/dpmc_$tag
done
# weak 3 of the round-5 review: the measured gap of the bf16 decoder operands against the PURE fp32 oracle, per item count
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "test_forward_parity" > $O/forward_parity.log 2>&1
grep -E "PURE fp32|max rel err probs|passed|failed" $O/forward_parity.log | head -40
ls -la $O/
