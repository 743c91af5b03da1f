# the NDCG gates under the three arithmetic / tower variants: how much of the curve statistic is noise
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gate
mkdir -p $O
for v in "fp32:LTGAN_D_ARITH=fp32" "x6_fused:LTGAN_D_ARITH=bf16x6" "x6_three:LTGAN_D_ARITH=bf16x6 LTGAN_TUNING=1048576" "x4_fused:LTGAN_D_ARITH=bf16x4"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 900 python -m pytest tests/test_gpu_ndcg_gate.py -q -m gpu -s -k "cpu_restatement" > $O/$name.log 2>&1
  echo "== $name"; grep -E "NDCG@100 after|curve, worst|passed|failed" $O/$name.log
done
