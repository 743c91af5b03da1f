import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import helpers as Hh
from oracle import ltg_oracle as O
from ltgan.engine import Engine, CsrRows, Pairs
import test_gpu_parity as T
I, B, n_batches, period = 9000, 48, 7, 3
rng = np.random.default_rng(4242)
P = O.init_generator(I, seed=3)
Xs = [Hh.random_history(rng, B, I, mean_nnz=14) for _ in range(n_batches)]
fakes = [T._fake_pairs(rng, X, I) for X in Xs]
engs = []
for lazy in (False, True):
    eng = Engine(I, lr=1e-3, precision="fp32", seed=1234, lazy_q0=lazy, q0_period=period)
    eng.set_generator(Hh.gen_to_engine(P))
    eng.q0_defer = True
    engs.append(eng)
dev = engs[0].device
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
acts = [e.new_acts(B) for e in engs]
for s in range(8):
    X = Xs[s % n_batches]
    rows, gen, pop = fakes[s % n_batches]
    slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
    batch = CsrRows(t(X.indptr.astype(np.int32)), t(X.indices.astype(np.int32)), 0, B, uptr=t(uptr), rowidx=t(rowidx), csr_pos=t(pos), n_unique=nu)
    fake = Pairs(t(pop), t(gen), t(rows))
    cnt = torch.tensor([int(((gen >= 0) & (pop >= 0)).sum())], dtype=torch.int32, device=dev)
    ls = [e.g_step(batch, fake, a, cnt, 0.05, rng_step=10 + s, d_rng_step=200 + s).clone() for e, a in zip(engs, acts)]
    torch.cuda.synchronize()
    print("step", s, "loss eq", torch.equal(ls[0], ls[1]), "h1 eq", torch.equal(acts[0].h1, acts[1].h1))
    # compare touched rows of W after the step
    items = np.unique(X.indices)
    it = torch.from_numpy(items).to(dev).long()
    for name, a, b in (("W", engs[0].g_p[0], engs[1].g_p[0]), ("m", engs[0].g_m[0], engs[1].g_m[0]), ("v", engs[0].g_v[0], engs[1].g_v[0])):
        d = (a[it] != b[it]).any(dim=1)
        print("   touched rows differing in", name, int(d.sum()), "of", len(items), [(int(items[k]), int(engs[1].q0_last[items[k]])) for k in torch.nonzero(d)[:4, 0].tolist()])
    for i in (1, 2, 4):
        print("   p%d eq" % i, torch.equal(engs[0].g_p[i], engs[1].g_p[i]))
