set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp8_backward" 2>&1 | grep -E "^E  |passed|failed" | head -20
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ltgan
from ltgan.engine import Engine, Pairs
hs=(2048,1024,512,256); I=500; nr,nf=900,921
rng=np.random.default_rng(77)
a=Engine(I,h_sizes=hs,lr=1e-3,precision="fp32",seed=1234,d_precision="fp8"); b=Engine(I,h_sizes=hs,lr=1e-3,precision="fp32",seed=1234,d_precision="fp8")
b.cfg.reserved0=1<<19
b.set_discriminator(a.d_emb.cpu().numpy(), [p.cpu().numpy() for p in a.d_p])
dev=a.device
t=lambda x: torch.from_numpy(x.astype(np.int32)).to(dev)
rp,rn,fp,fn=(rng.integers(0,I,k) for k in (nr,nr,nf,nf))
real,fake=Pairs(t(rp),t(rn)),Pairs(t(fp),t(fn))
n=nr+nf
names=["w1","b1","w2","b2","w3","b3","w4","b4"]
for lo,hi in ((0,n),(607,1366),(0,900),(900,n),(0,128),(0,1024),(0,1025)):
    ga=torch.empty(a.d_grad_floats(),dtype=torch.float32,device=dev); gb=torch.empty_like(ga)
    a.d_grad(real,fake,lo,hi,ga,keep_prob=0.7,rng_step=11); b.d_grad(real,fake,lo,hi,gb,keep_prob=0.7,rng_step=11)
    torch.cuda.synchronize()
    off=0; out=[]
    for nm,p in zip(names,a.d_p):
        k=p.numel(); x,y=ga[off:off+k].double(),gb[off:off+k].double(); off+=k
        out.append("%s %.1e" % (nm,(x-y).norm().item()/max(y.norm().item(),1e-30)))
    print(lo,hi," ".join(out), "loss", float(ga[off]), float(gb[off]))
PY
