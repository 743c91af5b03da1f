set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_run3
mkdir -p $O
for f in 0 0 1 2 3; do
LTGAN_TEST_PIPE_FLAGS=$f timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_adam and bf16" 2>&1 | grep -E "AssertionError: |passed|failed" | tr '\n' ' '
echo " <- flags $f"
done
MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tests/dist_shard_worker.py c4 200 bf16 > $O/c4_worker.log 2>&1
grep -v "Warning\|warn" $O/c4_worker.log | grep -B2 -A12 "Traceback\|Error" | head -60
