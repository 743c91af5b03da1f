# the bench lines again, now that profiles/ holds round 6's rocprofv3 summaries and PMC traffic (roofline.frac follows from THEM)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_evidence
mkdir -p $O
python bench.py --steps 20 --warmup 5 2>$O/bench_askubuntu.err | tail -1 > $O/r6_bench_askubuntu.json
python bench.py --workload c4 --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_ml20m.json
python bench.py --workload custom:25024 --parallelism item-shard --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_mid25k_item_shard.json
python bench.py --d-sizes 2048,1024,512,256 --d-precision fp8 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_askubuntu_wide_fp8.json
python bench.py --d-arith fp32 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_askubuntu_d_arith_fp32.json
python bench.py --workload ml20m --users 136000 --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/r6_bench_ml20m_full_136k_users.json
ls -la $O/*.json
