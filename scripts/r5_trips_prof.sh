cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_trips_prof
mkdir -p $O
for kv in "new=" "base=$R/ab_live/libltg_base.so"; do
  name=${kv%%=*}; lib=${kv#*=}
  for wl in "mid25k:--workload custom:25024 --parallelism item-shard" "ml20m:--workload ml20m --users 3200" "ask:"; do
    w=${wl%%:*}; extra=${wl#*:}
    LTG_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/bench.py $extra --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $O/log_${name}_$w.txt 2>&1
    f=$(find $O/p -name "*kernel_stats.csv" | head -1); cp "$f" $O/${name}_${w}_kernel_stats.csv; rm -rf $O/p
  done
done
python3 - <<'PY'
import csv, os, re
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5_trips_prof"
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]; n = n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n[:28]
        d[n] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    return d
for w in ("mid25k", "ml20m", "ask"):
    a, b = load("%s/new_%s_kernel_stats.csv" % (O, w)), load("%s/base_%s_kernel_stats.csv" % (O, w))
    print("==", w, "(avg us: base -> new)")
    for k in sorted(b, key=lambda k: -b[k][0] * b[k][1])[:22]:
        if k in a: print("  %-30s %5d  %8.2f -> %8.2f" % (k[:30], b[k][1], b[k][0], a[k][0]))
PY
