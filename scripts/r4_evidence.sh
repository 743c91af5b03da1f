# Round-4 FINAL evidence, one box: parity suite, bench JSON lines, kernel traces, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, no
# trace domains) and the SQ counter pass.  Everything lands in gpurun_out/r4_evidence/ (copied into profiles/ by hand).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4_evidence
mkdir -p $O
python bench.py --steps 20 --warmup 5 2>$O/bench_askubuntu.err | tail -1 > $O/r4_bench_askubuntu.json
python bench.py --workload c4 --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_ml20m.json
python bench.py --workload custom:25024 --parallelism item-shard --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_mid25k_item_shard.json
python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_askubuntu_wide_fp8.json
python bench.py --workload ml20m --users 136000 --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_ml20m_full_136k_users.json
python bench.py --workload c4 --users 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r4_bench_c4_full_1m_users.json
python scripts/host_bound_probe.py 25024 2>/dev/null | grep "items=" > $O/r4_host_issue_mid25k.txt
for wl in "askubuntu:" "c4_3200users:--workload c4 --users 3200" "ml20m_3200users:--workload ml20m --users 3200" "mid25k:--workload custom:25024 --parallelism item-shard" "askubuntu_wide_fp8:--d-sizes 2048,1024,512,256 --d-precision fp8"; do
  name=${wl%%:*}; extra=${wl#*:}
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -- python3 $R/bench.py $extra --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_$name.log 2>&1
  cd $R
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/r4_${name}_kernel_stats.csv; rm -rf $O/prof_$name
done
# timeline of two steps of the one-call sharded step (per-rank proxy) and of the Askubuntu_Sample G step
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_mid -- python3 $R/bench.py --workload custom:25024 --parallelism item-shard --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_mid.log 2>&1
cd $R
f=$(find $O/trace_mid -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_enc0_fwd 2 > $O/r4_mid25k_timeline.txt; rm -rf $O/trace_mid
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_ask -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace_ask.log 2>&1
cd $R
f=$(find $O/trace_ask -name "*kernel_trace.csv" | head -1); python profiles/make_timeline.py "$f" fk_enc0_fwd 3 > $O/r4_askubuntu_g_step_timeline.txt; python profiles/make_timeline.py "$f" fk_d_l1 3 > $O/r4_d_step_timeline.txt; rm -rf $O/trace_ask
for wl in "askubuntu:" "c4:--workload c4 --users 1600"; do
  name=${wl%%:*}; extra=${wl#*:}
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    tag=${c%% *}; [ "$tag" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && tag=SQ
    cd /tmp
    rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_${name}_$tag -- python3 $R/bench.py $extra --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/pmc_${name}_$tag.log 2>&1
    cd $R
    f=$(find $O/pmc_${name}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$O/r4_${name}_$tag.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
    rm -rf $O/pmc_${name}_$tag
  done
done
ls -la $O/
