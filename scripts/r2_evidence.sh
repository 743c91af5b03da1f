# Round-2 evidence, one box: bench JSON lines, kernel traces, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and the
# SQ counter pass for Askubuntu_Sample and the C4-shaped workload.  Everything lands in gpurun_out/r2_*.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 20 --warmup 5 2>gpurun_out/r2_bench_askubuntu.err | tail -1 > gpurun_out/r2_bench_askubuntu.json
python bench.py --workload c4 --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r2_bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r2_bench_ml20m.json
for wl in "askubuntu:" "c4:--workload c4 --users 3200" "ml20m:--workload ml20m --users 3200"; do
  name=${wl%%:*}; extra=${wl#*:}
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py $extra --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/gpurun_out/prof_$name.log 2>&1
  cd $R
  f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_${name}_kernel_stats.csv; rm -rf gpurun_out/prof_$name
done
for wl in "askubuntu:" "c4:--workload c4 --users 1600"; do
  name=${wl%%:*}; extra=${wl#*:}
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    tag=${c%% *}; [ "$tag" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && tag=SQ
    cd /tmp
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${name}_$tag -- python3 $R/bench.py $extra --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/gpurun_out/pmc_${name}_$tag.log 2>&1
    cd $R
    f=$(find gpurun_out/pmc_${name}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "gpurun_out/r2_${name}_$tag.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
    rm -rf gpurun_out/pmc_${name}_$tag
  done
done
ls -la gpurun_out/r2_*
