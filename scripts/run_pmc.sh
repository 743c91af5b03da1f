# rocprofv3 PMC passes (separate runs per counter, no trace domains) for the traffic column of the roofline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in "askubuntu:" "c4:--users 1600"; do
  name=${wl%%:*}; extra=${wl#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${name}_$c -- python3 $R/bench.py --workload $name $extra --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline > $R/gpurun_out/pmc_${name}_$c.log 2>&1
    f=$(find $R/gpurun_out/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$R/gpurun_out/pmc_${name}_$c.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
    rm -rf $R/gpurun_out/pmc_${name}_$c
  done
done
ls -la $R/gpurun_out/*.csv
