set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_run4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam or g_step_parity or d_step" 2>&1 | tail -5 > $O/tests_parity.log
tail -3 $O/tests_parity.log
timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -25 > $O/tests_sharded.log
tail -3 $O/tests_sharded.log
cd /tmp
LTGAN_PIPE_FLAGS=$((160*256)) rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --workload custom:25024 --parallelism item-shard --warm-moments --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/trace.log 2>&1
cd $R
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/trace_tail.csv <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-1200:]
t0 = int(tail[0]["Start_Timestamp"])
print("kernel,queue,start_us,end_us,dur_us,grid,wg")
for r in tail:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print("%s,%s,%.2f,%.2f,%.2f,%s,%s" % (name, r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
PY
rm -rf $O/trace
head -3 $O/trace_tail.csv; wc -l $O/trace_tail.csv
