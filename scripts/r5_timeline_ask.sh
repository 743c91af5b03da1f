# round 5, final build: kernel timelines of the headline workload's D step and G step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_tl
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace_ask -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline --no-other-workloads > $O/trace_ask.log 2>&1
cd $R
f=$(find $O/trace_ask -name "*kernel_trace.csv" | head -1)
python profiles/make_timeline.py "$f" fk_enc0_fwd 3 > $O/r5_askubuntu_g_step_timeline.txt
python profiles/make_timeline.py "$f" fk_d_l1 3 > $O/r5_d_step_timeline.txt
rm -rf $O/trace_ask
head -30 $O/r5_d_step_timeline.txt
