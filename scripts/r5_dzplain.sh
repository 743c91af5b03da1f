# round 5, experiment 23: one-call step without a communicator: the slab sum applies the tanh derivative, dz is the plain fk_dz
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_dzplain
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "pipelined or one_call or lazy or g_step or sharded or rccl or world" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= predz=$GRAFT_REPO_ROOT/ab_live/libltg_predz.so"
{
echo "== ml20m"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== c4"; bash scripts/ab_libs.sh "$L" --workload c4 --users 3200
} 2>&1 | tee $O/ab.txt
