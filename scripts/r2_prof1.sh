# one kernel trace: bash scripts/r2_prof1.sh <tag> <bench args...>   -> gpurun_out/r2_<tag>_kernel_stats.csv
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_${tag}_kernel_stats.csv; rm -rf gpurun_out/prof_$tag
python3 - gpurun_out/r2_${tag}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:32]:
    print("%-60s %6s %10.1f us  %5s%%" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
