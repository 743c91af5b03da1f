"""Debug aid: after every G step, which tensors of the one-call step (ltg_g_step_sharded, no communicator) differ in bits from
ltg_g_step's?  Same setup as tests/test_gpu_parity.py::test_lazy_adam_clock_..."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ltgan  # noqa
import helpers as Hh
from oracle import ltg_oracle as O
from ltgan.engine import CsrRows, Engine, Pairs, Pipe
from test_gpu_parity import _fake_pairs

I, B, n_batches, period = 9000, 48, 7, 5
rng = np.random.default_rng(4242)
P = O.init_generator(I, seed=3)
Xs = [Hh.random_history(rng, B, I, mean_nnz=14) for _ in range(n_batches)]
fakes = [_fake_pairs(rng, X, I) for X in Xs]
engs, state = {}, {}
flags = int(os.environ.get("LTGAN_TEST_PIPE_FLAGS", "0"))
for variant in ("lazy", "one-call"):
    eng = Engine(I, lr=1e-3, precision="bf16", seed=1234, lazy_q0=True, q0_period=period)
    eng.set_generator(Hh.gen_to_engine(P))
    eng.q0_defer = True
    engs[variant] = (eng, eng.new_acts(B), Pipe(eng, B, flags=flags) if variant == "one-call" else None)
names = ["Wq0", "Wq1", "Wp0", "Wp1t", "bq0", "bq1", "bp0", "bp1"]
for s in range(8):
    X = Xs[s % n_batches]; rows, gen, pop = fakes[s % n_batches]
    snap = {}
    for variant, (eng, acts, pipe) in engs.items():
        dev = eng.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
        batch = CsrRows(t(X.indptr.astype(np.int32)), t(X.indices.astype(np.int32)), 0, B, uptr=t(uptr), rowidx=t(rowidx), csr_pos=t(pos), n_unique=nu)
        fake = Pairs(t(pop), t(gen), t(rows))
        cnt = torch.tensor([int(((gen >= 0) & (pop >= 0)).sum())], dtype=torch.int32, device=dev)
        if pipe is not None:
            go = eng.g_opts(cnt, 0.05, rng_step=10 + s, d_rng_step=200 + s)
            loss = eng.g_step_sharded(batch, fake, acts, go, pipe).clone()
            eng.pipe_join(pipe)
        else:
            loss = eng.g_step(batch, fake, acts, cnt, 0.05, rng_step=10 + s, d_rng_step=200 + s).clone()
        torch.cuda.synchronize()
        ws = eng.workspace(B, fake.n)
        snap[variant] = dict(loss=loss.cpu(), h1=acts.h1[:B].cpu(), mulv=acts.mulv[:B].cpu(), z=acts.z[:B].cpu(), h2=acts.h2[:B].cpu(), logits=acts.logits[:B].cpu(),
                             lse=acts.lse[:B].cpu(), kl=acts.kl_rows[:B].cpu(), q0_last=eng.q0_last.cpu(),
                             **{"p_" + n: x.cpu() for n, x in zip(names, eng.g_p)}, **{"m_" + n: x.cpu() for n, x in zip(names, eng.g_m)},
                             **{"v_" + n: x.cpu() for n, x in zip(names, eng.g_v)})
    a, b = snap["lazy"], snap["one-call"]
    bad = []
    for k in a:
        if not torch.equal(a[k], b[k]):
            d = (a[k] != b[k])
            where = d.nonzero()[:3].tolist()
            bad.append("%s (%d of %d differ, first %s, max abs %.3g)" % (k, int(d.sum()), d.numel(), where, float((a[k].double() - b[k].double()).abs().max())))
    print("step %d:" % s, "identical" if not bad else "; ".join(bad), flush=True)
