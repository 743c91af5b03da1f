# round 6: wall-clock stamps inside fk_d_bwd1 job A (id 3) and fk_d_l2 (id 2), exact fp32 MFMA against the six-term split: which phase moved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_stamps
mkdir -p $O
for id in 3 2; do
  for ar in fp32 bf16x6; do
    n=$([ $id = 3 ] && echo 760 || echo 580)
    echo "== stamp id $id d_arith $ar" | tee -a $O/stamps.txt
    LTGAN_D_ARITH=$ar LTGAN_D_FORK=1 LTG_HIP_LIB=$GRAFT_REPO_ROOT/ab_live/libltg_stamp$id.so python scripts/stamp_probe.py $n 2>/dev/null | grep -v "^{" | tee -a $O/stamps.txt
  done
done
