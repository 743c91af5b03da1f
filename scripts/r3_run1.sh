# round 3, GPU run 1: parity of the one-call sharded step, then same-box A/B of the 25 024-item per-rank proxy
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
O=gpurun_out/r3_run1
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "g_step_parity or lazy_adam or forward_parity or without_slot" 2>&1 | tail -15 > $O/tests_parity.log
tail -5 $O/tests_parity.log
B="python bench.py --workload custom:25024 --parallelism item-shard --warm-moments --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe"
P=$R/build_ab/libltg_prev.so
IE=$R/build_ab/libltg_ieee.so
run() { # name, env...
  n=$1; shift
  env "$@" $B 2>$O/$n.err | tail -1 > $O/$n.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); nb=d["config"]["batches"]; S=d["config"]["sub_epochs"]
    print("$n", round(d["value"]), {k: round(v,2) for k,v in d["phases_ms"].items()}, "g_step_us %.1f d_step_us %.1f" % (d["phases_ms"]["t_g"]*1e3/(nb*S), d["phases_ms"]["t_d"]*1e3/(nb*S)))
except Exception as e:
    print("$n failed", e)
PY
}
for rep in 1 2; do
run prev_$rep LTG_HIP_LIB=$P LTGAN_SHARDED_STEP=0 LTGAN_PIPE_STEP=0
run new_cut_$rep LTGAN_SHARDED_STEP=0
run new_pipe_$rep X=1
run new_pipe_ieee_$rep LTG_HIP_LIB=$IE
run new_pipe_f1_$rep LTGAN_PIPE_FLAGS=1
run new_pipe_f2_$rep LTGAN_PIPE_FLAGS=2
run new_pipe_f4_$rep LTGAN_PIPE_FLAGS=4
run new_pipe_f3_$rep LTGAN_PIPE_FLAGS=3
done
python scripts/host_bound_probe.py 25024 > $O/host_probe.log 2>&1; tail -3 $O/host_probe.log
# single GPU, C3-shaped (20 000 items) and Askubuntu: pipelined one-call step / fast Adam arithmetic vs the previous build
B="python bench.py --workload ml20m --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-probe"
run c3_prev LTG_HIP_LIB=$P LTGAN_PIPE_STEP=0
run c3_new_nopipe LTGAN_PIPE_STEP=0
run c3_new_pipe X=1
run c3_prev2 LTG_HIP_LIB=$P LTGAN_PIPE_STEP=0
run c3_new_pipe2 X=1
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-probe"
run ask_prev LTG_HIP_LIB=$P
run ask_new X=1
run ask_ieee LTG_HIP_LIB=$IE
run ask_prev2 LTG_HIP_LIB=$P
run ask_new2 X=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_pipe -- python3 $R/bench.py --workload custom:25024 --parallelism item-shard --warm-moments --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $R/$O/prof_pipe.log 2>&1
cd $R
f=$(find $O/prof_pipe -name "*kernel_stats.csv" | head -1); cp "$f" $O/r3_mid25k_pipe_kernel_stats.csv; rm -rf $O/prof_pipe
head -30 $O/r3_mid25k_pipe_kernel_stats.csv | cut -c1-150
