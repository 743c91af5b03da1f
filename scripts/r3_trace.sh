# kernel-trace timeline of the last steps of a bench run: scripts/r3_trace.sh <out name> <bench args...>
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_trace
mkdir -p $O
name=$1; shift
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $O/trace_$name.log 2>&1
f=$(find $O/trace_$name -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/${name}_tail.csv <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", ";")
# a window in the D phase and one at the end (G phase)
n = len(rows)
for a, b in ((n // 2 - 300, n // 2 - 240), (n - 80, n)):
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        print("%s,%s,%.2f,%.2f,%.2f,%s" % (nm(r), r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
    print("----")
PY
rm -rf $O/trace_$name
