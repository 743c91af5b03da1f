# round 5, first box: the new tests (world size 8 on the gloo-on-one-GPU rig, warm moments at 200 000 items, span forward, sampler at 20 000
# items, the config-5 gate with the discriminator's own curves) + this box's numbers for the three large-slab workloads before any kernel change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_check1
mkdir -p $O
timeout 3000 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "eight_rank" --durations=0 > $O/pytest_eight.log 2>&1; echo "eight rc=$?" >> $O/pytest_eight.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ndcg_gate.py -q -m gpu -s -k "warm_moments or span_of_batches or 20000_items or wide_fp8" --durations=0 > $O/pytest_new.log 2>&1; echo "new rc=$?" >> $O/pytest_new.log
python bench.py --workload c4 --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --workload ml20m --users 6400 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_ml20m.json
python bench.py --workload custom:25024 --parallelism item-shard --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_mid25k.json
tail -5 $O/pytest_eight.log $O/pytest_new.log
