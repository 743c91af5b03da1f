set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_askubuntu.json
python bench.py --workload c4 --users 6400 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ml20m.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ask -- python3 $R/bench.py --steps 1 --warmup 1 --no-probe --no-cpu-baseline > $R/gpurun_out/prof_ask.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/bench.py --workload c4 --users 3200 --steps 1 --warmup 1 --no-probe --no-cpu-baseline > $R/gpurun_out/prof_c4.log 2>&1
cd $R
find gpurun_out/prof_ask gpurun_out/prof_c4 -name "*kernel_stats.csv" | head
