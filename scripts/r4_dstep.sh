# round 4, discriminator step (Askubuntu_Sample): (a) its launch structure -- the same five grids with their registers and LDS, returning at
# once (build_ab/libltg_dempty.so, -DLTG_D_EMPTY); (b) jobs B / C of backward stage 1 riding in stage 2's launch (--variant 32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-other-workloads --no-probe"
run() {  # name lib -- args
  name=$1; lib=$2; shift 2
  LTG_HIP_LIB=$lib python bench.py $B "$@" 2>/dev/null | tail -1 > $O/ab_tmp.json
  python - $O/ab_tmp.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
nb = d["config"]["batches"]
print("AB %-22s users/s %7d  d_step_us %5.1f  g_step_us %6.1f  phases %s" % (sys.argv[2], round(d["value"]), d["phases_ms"]["t_d"] * 1e3 / (nb * 10), d["phases_ms"]["t_g"] * 1e3 / (nb * 10), {k: round(v, 1) for k, v in d["phases_ms"].items()}))
PY
}
for rep in 1 2 3; do
  run shipped ""
  run bc_in_stage2 "" --variant 32
  run empty_d_kernels $R/build_ab/libltg_dempty.so
done 2>&1 | grep "^AB" | sort -s -k2,2 | tee $O/d_step_floor.txt
for v in 0 32; do
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_d$v -- python3 $R/bench.py $B --variant $v --steps 1 --warmup 1 > $O/tr_d$v.log 2>&1
cd $R
f=$(find $O/tr_d$v -name "*kernel_stats.csv" | head -1)
echo "== variant $v"; grep -E "fk_d_|Name" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-60,200-260
rm -rf $O/tr_d$v
done
timeout 1500 python -m pytest tests -m gpu -q --timeout 1500 --deselect tests/test_gpu_ndcg_gate.py -k "not (forward_parity or g_step_parity or d_step_parity or test_d_step_cut)" 2>&1 | tail -8
