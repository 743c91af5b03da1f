# round 6: 32-bit / hoisted address arithmetic in the one-kernel tower -- same-box A/B (new = the committed build, idx = the variant) + kernel averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_tower_idx
mkdir -p $O
LTG_HIP_LIB=$R/ab_live/libltg_idx.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "forward_only_tower" 2>&1 | tail -1
{ bash scripts/ab_libs.sh "new= idx=$R/ab_live/libltg_idx.so" --steps 10; } 2>&1 | tee $O/ab.txt
for v in new idx; do
  lib=""; [ $v = idx ] && lib=$R/ab_live/libltg_idx.so
  cd /tmp
  LTG_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_$v.log 2>&1
  cd $R
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a $O/ab.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fkt_d_tower" in r["Name"]: print(sys.argv[2], "fkt_d_tower calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3, 2))
PY
  rm -rf $O/prof_$v
done
