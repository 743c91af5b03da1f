# rocprofv3 PMC pass with SQ counters (MFMA busy, waits) -- no trace domains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in "askubuntu:" "c4:--users 1600"; do
  name=${wl%%:*}; extra=${wl#*:}
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_${name}_SQ -- python3 $R/bench.py --workload $name $extra --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline > $R/gpurun_out/pmc_${name}_SQ.log 2>&1
  f=$(find $R/gpurun_out/pmc_${name}_SQ -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$R/gpurun_out/pmc_${name}_SQ.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
  rm -rf $R/gpurun_out/pmc_${name}_SQ
done
ls -la $R/gpurun_out/pmc_*_SQ.csv
