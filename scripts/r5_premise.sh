# round 5: what could an earlier hand-over of the shadow gain at most?  Same-box builds: shipped; the streaming forward NOT waiting for the
# previous update's end (word 7; results wrong); no weight update at all (the caller's stream alone on the chip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_premise
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generic" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
L="new= now7=$GRAFT_REPO_ROOT/ab_live/libltg_now7.so noupd=$GRAFT_REPO_ROOT/ab_live/libltg_noupd.so"
echo "== ml20m (20 000 items, no communicator)"; bash scripts/ab_libs.sh "$L" --workload ml20m --users 6400
echo "== custom:25024 item-shard (RCCL at world size 1)"; bash scripts/ab_libs.sh "$L" --workload custom:25024 --parallelism item-shard
