# full default-config training run (config.ini defaults: NUM_EPOCH = 80 -> 10 sub-epochs) on Askubuntu_Sample + test.py
set -e
R=$GRAFT_REPO_ROOT
D=/tmp/ltgan_full
rm -rf $D && mkdir -p $D/run
python - <<PY
import sys; sys.path.insert(0, "$R")
from ltgan.dataset import materialize_askubuntu
materialize_askubuntu("$R/tests/golden/askubuntu_raw.npz", "$D/Askubuntu_Sample")
PY
cp $R/long-tail-gan_amd/config.ini $D/run/
cd $D/run
T0=$(date +%s.%N)
python $R/long-tail-gan_amd/train.py $D/Askubuntu_Sample > train.log 2> err.log || (tail -20 err.log; exit 1)
T1=$(date +%s.%N)
python -c "print(\"train.py wall seconds: %.1f\" % ($T1 - $T0))"
grep "Vad: NDCG" train.log | awk '{print NR-1, $7, $9, $11}' > ndcg_curve.txt
tail -3 ndcg_curve.txt
ck=$(ls -d chkpt/*)/model_79.pt
python $R/long-tail-gan_amd/test.py $D/Askubuntu_Sample $ck | tail -1
cp ndcg_curve.txt $R/gpurun_out/full_run_ndcg_curve.txt
grep -c "nan" train.log || true
