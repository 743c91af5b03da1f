"""GPU parity tests: the HIP path (through the C ABI, via ltgan.engine) against the CPU oracle on the
same seeded inputs.  Tolerances: fp32 path 2e-4 relative to the tensor's max magnitude (fp32
accumulation order differs from the fp64 oracle); bf16 path 1e-3 against the oracle fed the SAME
bf16-rounded decoder operands, which is north_star's 1e-3 relative bound.
"""
import os

import numpy as np
import pytest

import helpers as Hh
from oracle import ltg_oracle as O

pytestmark = pytest.mark.gpu

SEED = 1234


def _engine(I, precision, hs=(100, 150, 250, 300), lr=1e-4, **kw):
    import torch
    from ltgan.engine import Engine
    assert torch.cuda.is_available()
    return Engine(I, h_sizes=hs, lr=lr, precision=precision, seed=SEED, **kw)


def _problem(I, B, seed=0, mean_nnz=18):
    rng = np.random.default_rng(seed)
    X = Hh.random_history(rng, B, I, mean_nnz=mean_nnz)
    P = O.init_generator(I, seed=seed + 1)
    return rng, X, P


def _upload_batch(eng, X, with_csc=False):
    import torch
    from ltgan.engine import CsrRows
    dev = eng.device
    indptr = torch.from_numpy(X.indptr.astype(np.int32)).to(dev)
    indices = torch.from_numpy(X.indices.astype(np.int32)).to(dev)
    if with_csc:
        slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return CsrRows(indptr, indices, 0, X.shape[0], slot=up(slot), uptr=up(uptr), rowidx=up(rowidx), csr_pos=up(pos), n_unique=nu)
    return CsrRows(indptr, indices, 0, X.shape[0])


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 1e-3)])
@pytest.mark.parametrize("I,B", [(1000, 100), (777, 37), (64, 1), (2500, 130), (8200, 100), (8200, 150),
                                 (20000, 100), (200000, 100)])     # the last two: BASELINE configs 3 and 4 (item counts, BATCH_SIZE)
def test_forward_parity(precision, tol, I, B):
    import torch
    rng, X, P = _problem(I, B, seed=I + B)
    eng = _engine(I, precision)
    eng.set_generator(Hh.gen_to_engine(P))
    acts = eng.new_acts(B)
    batch = _upload_batch(eng, X)
    probs = torch.empty(B, I, dtype=torch.float32, device=eng.device)
    step, keep = 7, 0.75
    eng.forward(batch, acts, keep_prob=keep, is_training=1.0, rng_step=step, probs_out=probs)
    torch.cuda.synchronize()
    mask = Hh.dropout_mask_dense(SEED, step, B, I, keep)
    eps = Hh.eps_dense(SEED, step, B, eng.Z)
    F = O.vae_forward(P, X.toarray(), mask, keep, eps, 1.0, 1.0, np.float64, quant=(precision == "bf16"))
    got = {k: getattr(acts, k).cpu().numpy() for k in ("h1", "mulv", "z", "h2", "logits", "lse", "kl_rows")}
    assert Hh.rel_err(got["h1"], F["h1"]) < 2e-5
    assert Hh.rel_err(got["mulv"], np.concatenate([F["mu"], F["logvar"]], 1)) < 1e-4
    assert Hh.rel_err(got["z"], F["z"]) < 1e-4
    assert Hh.rel_err(got["h2"], F["h2"]) < 1e-4
    # KL terms 0.5 (-lv + e^lv + mu^2 - 1) cancel to ~lv^2 / 4 in fp32 (MultiVAE.py:161 as written, also in the reference): each of
    # the Z terms carries up to one fp32 ulp of e^lv ~ 1, which dominates when the activations are tiny (Xavier at I = 200 000)
    assert np.abs(got["kl_rows"] - F["KL_rows"]).max() < 1e-4 * np.abs(F["KL_rows"]).max() + 6e-8 * eng.Z
    assert Hh.rel_err(got["logits"], F["logits"]) < tol
    assert np.abs(got["lse"] - F["lse"]).max() < tol
    # probabilities: 1e-3 relative, element-wise (north_star)
    p = probs.cpu().numpy()
    perr = np.max(np.abs(p - F["probs"]) / F["probs"])
    print("I=%d B=%d %s: max rel err probs %.2e logits %.2e" % (I, B, precision, perr, Hh.rel_err(got["logits"], F["logits"])))
    assert perr < 1e-3
    # against the pure-fp32-operand oracle the bf16 path stays within 1e-2 on probabilities (element-wise)
    if precision == "bf16":
        F32 = O.vae_forward(P, X.toarray(), mask, keep, eps, 1.0, 1.0, np.float64, quant=False)
        gap = np.max(np.abs(p - F32["probs"]) / F32["probs"])
        print("I=%d B=%d bf16 operands against the PURE fp32 oracle: max rel gap probs %.2e, logits %.2e (DESIGN.md section 6 quotes these)" %
              (I, B, gap, Hh.rel_err(got["logits"], F32["logits"])))
        assert gap < 1e-2          # (measured 5.3e-4 .. 5.0e-3 over the eight shapes: profiles/r6_bf16_vs_pure_fp32_gap.txt; the bound was 2e-2 until round 6)


@pytest.mark.parametrize("I,rows", [(20000, (100, 100, 50)), (1000, (100, 100, 100, 1))])
def test_forward_parity_over_a_span_of_batches(I, rows):
    """Phase C's forward as the trainer issues it (train.py:192-200 over consecutive batches; `ltg_fwd_opts.rows_per_step`): ONE launch
    sequence over several batches, batch k drawing its input dropout with counter rng_step + k and its LOCAL row numbers -- against the
    oracle batch by batch (is_training = 0: no epsilon, as train.py:200 feeds it).  I = 20 000: BASELINE config 3's item count."""
    import scipy.sparse as sp
    import torch
    R, bs = sum(rows), rows[0]
    rng, X, P = _problem(I, R, seed=3 * I + R)
    eng = _engine(I, "bf16")
    eng.set_generator(Hh.gen_to_engine(P))
    acts = eng.new_acts(R)
    batch = _upload_batch(eng, X)
    probs = torch.empty(R, I, dtype=torch.float32, device=eng.device)
    step, keep = 21, 0.75
    eng.forward(batch, acts, keep_prob=keep, is_training=0.0, rng_step=step, probs_out=probs, rows_per_step=bs)
    torch.cuda.synchronize()
    got = {k: getattr(acts, k).cpu().numpy() for k in ("h1", "h2", "logits", "lse")}
    p = probs.cpu().numpy()
    r0 = 0
    for k, n in enumerate(rows):
        Xk = X[r0:r0 + n].toarray()
        mask = Hh.dropout_mask_dense(SEED, step + k, n, I, keep)
        F = O.vae_forward(P, Xk, mask, keep, np.zeros((n, eng.Z)), 0.0, 1.0, np.float64, quant=True)
        sl = slice(r0, r0 + n)
        assert Hh.rel_err(got["h1"][sl], F["h1"]) < 2e-5, k
        assert Hh.rel_err(got["h2"][sl], F["h2"]) < 1e-4, k
        assert Hh.rel_err(got["logits"][sl], F["logits"]) < 1e-3, k
        assert np.abs(got["lse"][sl] - F["lse"]).max() < 1e-3, k
        perr = np.max(np.abs(p[sl] - F["probs"]) / F["probs"])
        assert perr < 1e-3, (k, perr)
        r0 += n


@pytest.mark.parametrize("I,B", [(8200, 100), (25032, 128), (20000, 1), (65536, 100), (65544, 113)])
def test_forward_parity_of_the_second_streaming_form(I, B):
    """k_dec1_fwd_stream2 (h2 resident in LDS, every wave its own 32-item tiles; the library uses it from 65 536 items) forced at every
    streaming size through tuning-knob bit 17: ragged slabs (I % 32 = 8), 8 batch tiles (B > 112), a single row, fewer tiles than waves --
    logits, lse and probabilities against the oracle exactly as test_forward_parity states them, and against the first form (bit 26)."""
    import torch
    rng, X, P = _problem(I, B, seed=I + B)
    outs = {}
    for knob in (1 << 17, 1 << 26):
        eng = _engine(I, "bf16")
        eng.cfg.tuning = knob
        eng.set_generator(Hh.gen_to_engine(P))
        acts = eng.new_acts(B)
        batch = _upload_batch(eng, X)
        probs = torch.empty(B, I, dtype=torch.float32, device=eng.device)
        eng.forward(batch, acts, keep_prob=0.75, is_training=1.0, rng_step=7, probs_out=probs)
        torch.cuda.synchronize()
        outs[knob] = (acts.logits[:B].cpu().numpy().copy(), acts.lse[:B].cpu().numpy().copy(), probs.cpu().numpy())
        del eng
    mask = Hh.dropout_mask_dense(SEED, 7, B, I, 0.75)
    eps = Hh.eps_dense(SEED, 7, B, 200)
    F = O.vae_forward(P, X.toarray(), mask, 0.75, eps, 1.0, 1.0, np.float64, quant=True)
    lg, lse, p = outs[1 << 17]
    assert Hh.rel_err(lg, F["logits"]) < 1e-3
    assert np.abs(lse - F["lse"]).max() < 1e-3
    assert np.max(np.abs(p - F["probs"]) / F["probs"]) < 1e-3
    lg1, lse1, p1 = outs[1 << 26]
    assert Hh.rel_err(lg, lg1) < 1e-5 and np.abs(lse - lse1).max() < 1e-5      # (same K order per logit; other tiles, other lanes)


def _check_adam_move(move_got, move_want, m_want, lr_t, tag):
    """First-step Adam moves are lr_t*0.1g/(0.0316|g|+1e-8): a sign-like function of g, so rounding
    noise on a near-cancelling gradient (|g| << its terms) is amplified without bound.  Elements
    whose gradient is at least 2% of the tensor's largest are compared strictly (5% of lr_t); the
    rest only by the bound |move| <= lr_t*(1-b1)/sqrt(1-b2)."""
    diff = np.abs(np.asarray(move_got, np.float64) - move_want)
    big = np.abs(m_want) > 0.02 * np.abs(m_want).max()
    if big.any():
        assert diff[big].max() < 0.05 * lr_t + 1e-7, tag
    assert np.abs(move_got).max() < 3.2 * lr_t, tag


def _fake_pairs(rng, X, I, per_user=5):
    """random (row, gen, pop) triples with a few holes, sorted by row like the sampler's slots"""
    B = X.shape[0]
    rows, gen, pop = [], [], []
    for b in range(B):
        if rng.random() < 0.1:
            continue
        k = int(rng.integers(1, per_user + 1))
        g = np.sort(rng.choice(I, size=k, replace=False))
        for gi in g:
            rows.append(b)
            if rng.random() < 0.05:
                gen.append(-1)
                pop.append(-1)
            else:
                gen.append(int(gi))
                pop.append(int(rng.integers(0, I)))
    return np.array(rows, np.int32), np.array(gen, np.int32), np.array(pop, np.int32)


def _warm_moments(shapes_like, grads, seed):
    """Adam moments as after many updates: m ~ N(0, rms(g)), v ~ U(0.25, 4) rms(g)^2 per tensor (fp32-representable), so that the
    step's quotient m / (sqrt(v) + eps) is a smooth function of every input -- what a first step from zero moments (a sign
    function of g) cannot test."""
    r = np.random.default_rng(seed)
    m0, v0 = {}, {}
    for k, ref in shapes_like.items():
        g = np.asarray(grads[k], np.float64)
        nz = g[g != 0]
        s = float(np.sqrt(np.mean(nz * nz))) if nz.size else 1e-3
        m0[k] = (r.normal(0.0, s, ref.shape)).astype(np.float32)
        v0[k] = (r.uniform(0.25, 4.0, ref.shape) * s * s).astype(np.float32)
    return m0, v0


def _check_warm_adam(p_got, p0, p_want, m_got, m_want, v_got, v_want, tol, tag, t, lr=1e-3, eps=1e-8, max_norm=False):
    """theta-move, m and v ELEMENT-WISE against TF-Adam in fp64 (train.py:160-164) from injected non-zero moments: pins the hardware
    sqrt / reciprocal form of the library's adam_move to m / (sqrt(v) + eps).
    (a) the quotient alone: every element's move against lr_t m / (sqrt(v) + eps) evaluated in fp64 from the moments the DEVICE
        wrote -- no gradient error in it, so the bound is a few fp32 ulp (v_sqrt_f32 and v_rcp_f32 are 1 ulp each);
    (b) end to end against the oracle's update: element-wise (1e-4) where the move is not negligible on the fp32 path; in the max norm
        (1e-3, north_star's bound) with bf16 decoder operands."""
    p0 = np.asarray(p0, np.float64)
    mv_got, mv_want = np.asarray(p_got, np.float64) - p0, np.asarray(p_want, np.float64) - p0
    ulp = np.spacing(np.maximum(np.abs(p0), np.abs(np.asarray(p_got, np.float64))).astype(np.float32)).astype(np.float64)   # theta is rounded to fp32 once
    mg, vg = np.asarray(m_got, np.float64), np.asarray(v_got, np.float64)
    # lr_t as the library forms it (csrc make_adam): lr, beta1, beta2 are the fp32 values of ltg_config, the powers are taken in fp64
    # (TF keeps beta^t as an fp32 variable multiplied up step by step: either differs from the exact 0.999^t by ~1e-5 at t = 138,
    # 6e-6 on lr_t -- far inside (b)'s bound, but 3x this check's)
    lr32, b1, b2 = (float(np.float32(x)) for x in (lr, 0.9, 0.999))
    lr_t = np.float64(np.float32(lr32 * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)))
    q = -lr_t * mg / (np.sqrt(vg) + np.float64(np.float32(eps)))
    qerr = (np.abs(mv_got - q) - ulp) / np.maximum(np.abs(q), 1e-30)
    w = np.unravel_index(int(np.argmax(qerr)), qerr.shape)
    assert float(qerr[w]) < 2e-6, (tag, "quotient", float(qerr[w]), "at", w, "p0 %.9g p_got %.9g m %.9g v %.9g move %.9g want %.9g ulp %.3g" % (
        p0[w], np.asarray(p_got, np.float64)[w], mg[w], vg[w], mv_got[w], q[w], ulp[w]))
    err = np.abs(mv_got - mv_want)
    if max_norm:
        # bf16 decoder operands: dlogits are ROUNDED to bf16 before the two products that consume them, so an element of the gradient
        # differs from the (equally quantised) oracle's by whatever a value near a rounding boundary moves it -- a bound relative to
        # the tensor's largest element, like every bf16 comparison of this file; (a) above stays element-wise
        worst = float((err - ulp).max() / np.abs(mv_want).max())
        assert worst < tol, (tag, "theta move (max norm)", worst)
        assert Hh.rel_err(m_got, m_want) < tol and Hh.rel_err(v_got, v_want) < 2 * tol, (tag, "m / v")
        return worst
    sel = np.abs(mv_want) > 0.01 * np.abs(mv_want).max()
    worst = float(((err - ulp)[sel] / np.abs(mv_want)[sel]).max())
    assert worst < tol, (tag, "theta move", worst)
    selm = np.abs(m_want) > 0.01 * np.abs(m_want).max()
    assert float((np.abs(m_got - m_want)[selm] / np.abs(m_want)[selm]).max()) < tol, (tag, "m")
    assert float((np.abs(v_got - v_want) / v_want).max()) < tol, (tag, "v")
    return worst


G_CASES = [(1000, 100, "step"), (333, 17, "step"), (6000, 100, "step"), (6000, 100, "step-config-d"), (8200, 100, "step"),
           (8200, 150, "step"), (8200, 100, "one-call"), (20000, 100, "step"), (25024, 100, "step"),
           (25024, 100, "one-call-config-d"), (200000, 100, "step"), (200000, 100, "one-call"),
           # ragged slabs (I % 32 != 0): the last I % 32 item rows go through the streaming kernels' tail path
           (25032, 100, "one-call"), (200008, 100, "step"),
           # tuning-knob bit 18: the generic (round-1) kernels -- what sizes the latency path does not serve fall back to (z_dim % 4 != 0,
           # more than 256 rows, discriminator layers that are not multiples of 4) -- over the same oracle, small and streaming-sized slabs
           (1000, 100, "step-generic"), (333, 17, "step-generic"), (8200, 100, "step-generic")]


def _g_step_case(precision, I, B, path, warm):
    import torch
    from ltgan.engine import Pairs, Pipe
    rng, X, P = _problem(I, B, seed=11 * I + B)
    hs = (100, 150, 250, 300) if path.endswith("config-d") else (20, 24, 40, 36)
    D = O.init_discriminator(I, *hs, seed=5)
    rows, gen, pop = _fake_pairs(rng, X, I)
    valid = (gen >= 0) & (pop >= 0)
    cnt = int(valid.sum())
    step, dstep, keep, dkeep, anneal, lam = 3, 9, 0.75, 0.7, 0.13, 1.0
    t0 = 137 if warm else 4                 # the shared counter before this update (t = t0 + 1)
    # ---- oracle
    q = precision == "bf16"
    mask = Hh.dropout_mask_dense(SEED, step, B, I, keep)
    eps = Hh.eps_dense(SEED, step, B, 200)
    n = len(rows)
    dm = Hh.d_masks(SEED, dstep, n, hs[1:], dkeep)
    gid = np.where(valid, gen, 0)
    pid = np.where(valid, pop, 0)
    T = O.d_tower(D, pid, gid, dm, dkeep)
    sum_y = float((T["y"] * valid).sum())
    losses, g, F = O.g_loss_and_grads(P, X.toarray(), mask, keep, eps, anneal, lam, rows[valid], gen[valid], cnt, sum_y,
                                      1.0, np.float64, quant=q)
    ad = O.SharedAdam(1e-3)
    ad.t = t0
    m0 = v0 = None
    if warm:
        m0, v0 = _warm_moments({k: P[k] for k in O.G_KEYS}, g, seed=I + 7)
        for k in O.G_KEYS:
            ad.m[k], ad.v[k] = m0[k].astype(np.float64), v0[k].astype(np.float64)
    P64 = {k: np.asarray(v, np.float64) for k, v in P.items()}
    ad.apply(P64, g, O.G_KEYS)
    # ---- device
    eng = _engine(I, precision, hs=hs, lr=1e-3)
    assert eng.Z == 200
    if path.endswith("-generic"):
        eng.cfg.tuning = 1 << 18
    if warm:
        eng.set_generator(Hh.gen_to_engine(P), m=Hh.gen_to_engine(m0), v=Hh.gen_to_engine(v0))
    else:
        eng.set_generator(Hh.gen_to_engine(P))
    emb, darr = Hh.disc_to_engine(D)
    eng.set_discriminator(emb, darr)
    dev = eng.device
    fake = Pairs(torch.from_numpy(pop).to(dev), torch.from_numpy(gen).to(dev), torch.from_numpy(rows).to(dev))
    cnt_t = torch.tensor([cnt], dtype=torch.int32, device=dev)
    batch = _upload_batch(eng, X, with_csc=True)
    acts = eng.new_acts(B)
    eng.adam_t = t0  # exercise the shared counter
    if path.startswith("one-call"):
        assert eng.sharded_step_ok(B)
        pipe = Pipe(eng, B)
        go = eng.g_opts(cnt_t, anneal, lam, keep, 1.0, dkeep, step, dstep)
        loss = eng.g_step_sharded(batch, fake, acts, go, pipe)
        assert pipe.expired_waits() == 0
    else:
        loss = eng.g_step(batch, fake, acts, cnt_t, anneal, lam, keep, 1.0, dkeep, rng_step=step, d_rng_step=dstep)
    eng.g_flush()
    torch.cuda.synchronize()
    loss = loss.cpu().numpy()
    assert abs(loss[4] - sum_y) < 1e-4 * max(1.0, abs(sum_y))
    assert abs(loss[3] - losses["sum_p"]) < 2e-3 * abs(losses["sum_p"]) + 1e-7
    for i, k in enumerate(("g_loss", "vae_loss", "gan_loss")):
        assert abs(loss[i] - losses[k]) < 1e-3 * abs(losses[k]) + 1e-6, k
    # one TF-Adam update at the shared step t0 + 1
    want = Hh.gen_to_engine(P64)
    want_m = Hh.gen_to_engine({k: ad.m[k] for k in O.G_KEYS})
    want_v = Hh.gen_to_engine({k: ad.v[k] for k in O.G_KEYS})
    gtol = 2e-3 if q else 5e-4
    worst = 0.0
    for i in range(8):
        m_got = eng.g_m[i].cpu().numpy()
        v_got = eng.g_v[i].cpu().numpy()
        p_got = eng.g_p[i].cpu().numpy()
        if warm:
            worst = max(worst, _check_warm_adam(p_got, Hh.gen_to_engine(P)[i], want[i], m_got, want_m[i], v_got, want_v[i], 1e-3 if q else 1e-4, ("tensor", i), t0 + 1,
                                                max_norm=q))
            continue
        assert Hh.rel_err(m_got, want_m[i]) < gtol, ("m", i)
        assert Hh.rel_err(v_got, want_v[i]) < 2 * gtol, ("v", i)
        # theta moves by at most lr_t per element; compare the MOVE
        move_got = p_got - Hh.gen_to_engine(P)[i]
        move_want = want[i] - np.asarray(Hh.gen_to_engine(P)[i], np.float64)
        _check_adam_move(move_got, move_want, want_m[i], ad.lr_t(t0 + 1), ("theta", i))
    if warm:
        print("I=%d %s %s: worst relative error of a theta move from warm moments %.2e" % (I, precision, path, worst))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("I,B,path", G_CASES)
def test_g_step_parity(precision, I, B, path):
    """path "step": ltg_g_step.  "one-call": ltg_g_step_sharded without a communicator (the G step of every large item slab:
    bias + tanh and the tanh derivative folded into operand loaders, weight update and clock slice forked) -- same oracle, same
    tolerances.  I = 6 000: the middle-layer fast path without the streaming decoder kernels (4 096 < I < 8 192); I = 25 024: the
    slab one of eight ranks owns at 200 000 items.  "-config-d": config.ini's discriminator sizes inside the G step."""
    if I >= 20000 and precision != "bf16":
        pytest.skip("BASELINE configs 3 / 4 run the bf16 decoder path")
    if path.startswith("one-call") and precision != "bf16":
        pytest.skip("ltg_g_step_sharded serves the bf16 decoder path")
    _g_step_case(precision, I, B, path, warm=False)


@pytest.mark.parametrize("precision,I,B,path", [("fp32", 1000, 100, "step"), ("bf16", 1000, 100, "step"), ("fp32", 6000, 100, "step"),
                                                ("bf16", 8200, 100, "step"), ("bf16", 20000, 100, "step"), ("bf16", 25024, 100, "one-call"),
                                                ("bf16", 25032, 100, "one-call"),
                                                # BASELINE config 4's item count: the combination the driver's C4 bench leg runs (one call, warm
                                                # moments, 200 000 items), and the ragged slab through the five-launch step
                                                ("bf16", 200000, 100, "one-call"), ("bf16", 200008, 100, "step")])
def test_g_step_adam_quotient_from_warm_moments(precision, I, B, path):
    """The G step from injected NON-ZERO Adam moments at shared step t = 138 against oracle.SharedAdam (train.py:160-164): every
    theta move, m and v element-wise -- dense tile epilogues, the streaming weight update, the lazy clock's kernels (rows of W_q0
    the batch does not touch take a zero-gradient step) all go through the library's one adam_move."""
    _g_step_case(precision, I, B, path, warm=True)


@pytest.mark.parametrize("d_arith", ["fp32", "bf16x6"])
@pytest.mark.parametrize("nr,nf", [(900, 950), (33, 7), (1, 0), (0, 5), (260, 250)])
def test_d_step_parity(nr, nf, d_arith):
    """d_arith (ltg_config.d_arith): the exact fp32 MFMA and the six-term bf16 split of the same fp32 operands -- SAME bounds (the split leaves
    out 2^-26 per product, below fp32's own rounding); the wide sizes (260, 250) do not take the latency-path kernels: one arithmetic"""
    if d_arith != "fp32" and nr == 260:
        pytest.skip("config-5 sizes run the LDS-staged kernels: d_arith does not apply")
    _d_step_case(nr, nf, warm=False, d_arith=d_arith)


@pytest.mark.parametrize("nr,nf", [(900, 950), (33, 7)])
def test_d_step_parity_four_term_split(nr, nf):
    """d_arith = bf16x4 (opt-in, labelled NOT fp32-accurate: hi / mid terms only, 2^-17 per product): the same step against the same oracle at
    bounds 8x wider than fp32's (loss 1e-4 holds; first moments 4e-3, second 8e-3)"""
    _d_step_case(nr, nf, warm=False, d_arith="bf16x4", loosen=8.0)


@pytest.mark.parametrize("nr,nf", [(900, 950), (33, 7)])
def test_d_step_parity_of_the_generic_kernels(nr, nf):
    """tuning-knob bit 18: the discriminator step through the generic (round-1) kernels, same oracle and bounds"""
    _d_step_case(nr, nf, warm=False, knob=1 << 18)


@pytest.mark.parametrize("d_arith", ["fp32", "bf16x6"])
@pytest.mark.parametrize("nr,nf", [(900, 950), (33, 7), (260, 250)])
def test_d_step_adam_quotient_from_warm_moments(nr, nf, d_arith):
    """The D step from injected non-zero Adam moments at shared step t = 212 (train.py:160-163): the flat Adam sweep's theta move,
    m and v element-wise against oracle.SharedAdam."""
    if d_arith != "fp32" and nr == 260:
        pytest.skip("config-5 sizes run the LDS-staged kernels: d_arith does not apply")
    _d_step_case(nr, nf, warm=True, d_arith=d_arith)


def _d_step_case(nr, nf, warm, knob=0, d_arith=None, loosen=1.0):
    import torch
    from ltgan.engine import Pairs
    I = 500
    hs = (100, 150, 250, 300) if nr > 300 else (12, 20, 28, 16)
    if nr == 260:
        hs = (2048, 1024, 512, 256)          # BASELINE config 5: the wide discriminator (3 540 993 parameters), fp32
    rng = np.random.default_rng(nr * 7 + nf)
    D = O.init_discriminator(I, *hs, seed=3)
    D["b1"] = rng.normal(0, 0.05, D["b1"].shape).astype(np.float32)
    D["b3"] = rng.normal(0, 0.05, D["b3"].shape).astype(np.float32)
    D["b4"] = rng.normal(0, 0.05, D["b4"].shape).astype(np.float32)

    def mk(n):
        pop = rng.integers(0, I, n).astype(np.int32)
        nic = rng.integers(0, I, n).astype(np.int32)
        hole = rng.random(n) < 0.05
        pop[hole] = -1
        nic[hole] = -1
        return pop, nic

    rp, rn = mk(nr)
    fp, fn = mk(nf)
    step, keep = 21, 0.7
    t0 = 211 if warm else 2
    # oracle: masks are indexed by the logical row (real rows first, then fake rows)
    n = nr + nf
    dm = Hh.d_masks(SEED, step, n, hs[1:], keep)
    vr = rp >= 0
    vf = fp >= 0
    Tr = O.d_tower(D, np.where(vr, rp, 0), np.where(vr, rn, 0), [m[:nr] for m in dm], keep)
    Tf = O.d_tower(D, np.where(vf, fp, 0), np.where(vf, fn, 0), [m[nr:] for m in dm], keep)
    want_loss = -(np.log(Tr["y"]) * vr).sum() - (np.log(1 - Tf["y"]) * vf).sum()
    gr = O.d_tower_backward(D, Tr, [m[:nr] for m in dm], keep, -(1 - Tr["y"]) * vr)
    gf = O.d_tower_backward(D, Tf, [m[nr:] for m in dm], keep, Tf["y"] * vf)
    g = {k: gr[k] + gf[k] for k in gr}
    ad = O.SharedAdam(1e-3)
    ad.t = t0
    m0 = v0 = None
    if warm:
        m0, v0 = _warm_moments({k: np.asarray(D[k]) for k in O.D_KEYS}, {k: np.asarray(g[k]).reshape(np.asarray(D[k]).shape) for k in O.D_KEYS}, seed=nr + 3)
        for k in O.D_KEYS:
            ad.m[k], ad.v[k] = m0[k].astype(np.float64), v0[k].astype(np.float64)
    D64 = {k: np.asarray(v, np.float64) for k, v in D.items()}
    ad.apply(D64, g, O.D_KEYS)
    # device
    eng = _engine(I, "fp32", hs=hs, lr=1e-3, d_arith=d_arith)
    eng.cfg.tuning = knob
    emb, darr = Hh.disc_to_engine(D)
    if warm:
        eng.set_discriminator(emb, darr, m=[m0[k] for k in O.D_KEYS], v=[v0[k] for k in O.D_KEYS])
    else:
        eng.set_discriminator(emb, darr)
    dev = eng.device
    real = Pairs(torch.from_numpy(rp).to(dev), torch.from_numpy(rn).to(dev)) if nr else Pairs(torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), n=0)
    fake = Pairs(torch.from_numpy(fp).to(dev), torch.from_numpy(fn).to(dev)) if nf else Pairs(torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), n=0)
    eng.adam_t = t0
    loss = eng.d_step(real, fake, keep, rng_step=step)
    torch.cuda.synchronize()
    loss = float(loss.cpu().numpy()[0])
    assert abs(loss - want_loss) < 1e-4 * max(1.0, abs(want_loss))
    if warm:
        worst = 0.0
        for i, k in enumerate(O.D_KEYS):
            sh = np.asarray(D[k]).shape
            worst = max(worst, _check_warm_adam(eng.d_p[i].cpu().numpy().reshape(sh), np.asarray(D[k]), D64[k], eng.d_m[i].cpu().numpy().reshape(sh), ad.m[k],
                                                eng.d_v[i].cpu().numpy().reshape(sh), ad.v[k], 1e-4, ("d tensor", k), t0 + 1))
        print("D step (%d, %d): worst relative error of a theta move from warm moments %.2e" % (nr, nf, worst))
        return
    worst_m = worst_v = 0.0
    for i, k in enumerate(O.D_KEYS):
        m_got = eng.d_m[i].cpu().numpy().reshape(-1)
        worst_m = max(worst_m, Hh.rel_err(m_got, ad.m[k].reshape(-1)))
        assert Hh.rel_err(m_got, ad.m[k].reshape(-1)) < 5e-4 * loosen, ("m", k)
        v_got = eng.d_v[i].cpu().numpy().reshape(-1)
        worst_v = max(worst_v, Hh.rel_err(v_got, ad.v[k].reshape(-1)))
        assert Hh.rel_err(v_got, ad.v[k].reshape(-1)) < 1e-3 * loosen, ("v", k)
    print("D step (%d, %d) d_arith %s: loss rel err %.2e, worst first / second moment rel err %.2e / %.2e" %
          (nr, nf, eng.d_arith, abs(loss - want_loss) / max(1.0, abs(want_loss)), worst_m, worst_v))
    for i, k in enumerate(O.D_KEYS):
        move_got = eng.d_p[i].cpu().numpy().reshape(-1) - np.asarray(D[k], np.float64).reshape(-1)
        move_want = D64[k].reshape(-1) - np.asarray(D[k], np.float64).reshape(-1)
        _check_adam_move(move_got, move_want, ad.m[k].reshape(-1), ad.lr_t(3), ("theta", k))


@pytest.mark.parametrize("d_arith,tol", [("bf16x6", 2e-6), ("bf16x4", 2e-5), ("fp32", 2e-6)])
@pytest.mark.parametrize("hs,segs", [((100, 150, 250, 300), (1, 63, 64, 65, 700, 1311)), ((12, 20, 28, 16), (5, 130)), ((64, 96, 200, 320), (257,)),
                                     ((8, 5, 9, 3), (70,)), ((128, 33, 17, 7), (129,))])
def test_forward_only_tower_matches_oracle(hs, segs, d_arith, tol):
    """ltg_fake_tower_batched (discriminator.py:51-55 for many pair batches in one pass; consumed at train.py:155): y of every slot against the
    oracle's tower, segment by segment with the segment's own dropout counter -- through the ONE-kernel tower (csrc/ltg_tower.h: d_arith bf16x6 /
    bf16x4) and through the three-launch tower (d_arith fp32).  Ragged last row blocks, a one-row segment, holes (id -1 -> y = 0), layer sizes
    that are not multiples of 16 / 32, the largest fc layer the kernel takes (320)."""
    import torch
    from ltgan.engine import Pairs
    I, keep = 400, 0.7
    rng = np.random.default_rng(sum(segs) + hs[0])
    D = O.init_discriminator(I, *hs, seed=11)
    for k in ("b1", "b2", "b3", "b4"):
        D[k] = rng.normal(0, 0.05, D[k].shape).astype(np.float32)
    eng = _engine(I, "fp32", hs=hs, lr=1e-3, d_arith=d_arith)
    emb, darr = Hh.disc_to_engine(D)
    eng.set_discriminator(emb, darr)
    dev = eng.device
    n = sum(segs)
    pop, nic = rng.integers(0, I, n).astype(np.int32), rng.integers(0, I, n).astype(np.int32)
    hole = rng.random(n) < 0.04
    pop[hole] = -1
    nic[hole] = -1
    row0 = np.concatenate([[0], np.cumsum(segs)[:-1]]).astype(np.int32)
    seg_of = np.repeat(np.arange(len(segs), dtype=np.int32), segs)
    steps = (1000 + 7 * np.arange(len(segs))).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(dev)
    y = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
    eng.fake_tower_batched(Pairs(t(pop), t(nic)), t(seg_of), t(row0), t(steps), y, keep)
    torch.cuda.synchronize()
    got = y.cpu().numpy().astype(np.float64)
    worst = 0.0
    for sidx, (r0, ns) in enumerate(zip(row0, segs)):
        sl = slice(r0, r0 + ns)
        dm = Hh.d_masks(SEED, int(steps[sidx]), ns, hs[1:], keep)
        v = pop[sl] >= 0
        T = O.d_tower(D, np.where(v, pop[sl], 0), np.where(v, nic[sl], 0), dm, keep)
        want = np.where(v, T["y"], 0.0)
        worst = max(worst, float(np.max(np.abs(got[sl] - want))))
        assert np.all(got[sl][~v] == 0.0)
    print("forward-only tower %s d_arith %s: worst |y - oracle| %.2e over %d slots" % (hs, d_arith, worst, n))
    assert worst < tol, worst


def test_d_step_with_its_backward_jobs_on_the_aux_stream_is_bit_identical():
    """ltg_d_opts.aux_stream / sync (Engine.d_fork): jobs B / C of the backward's first stage (dw3, db3, dw4, db4, d_loss) beside the
    critical chain job A -> stage 2, handed over through device words -- the same kernels writing the same slab entries, so six steps
    from equal states must leave the same bits as the one-stream step: losses, every weight, every moment."""
    import torch
    from ltgan.engine import Pairs
    I, hs = 500, (100, 150, 250, 300)
    rng = np.random.default_rng(17)
    D = O.init_discriminator(I, *hs, seed=3)
    emb, darr = Hh.disc_to_engine(D)
    a, b = _engine(I, "fp32", hs=hs, lr=1e-3), _engine(I, "fp32", hs=hs, lr=1e-3)
    b.d_fork = False
    dev = a.device
    for e in (a, b):
        e.set_discriminator(emb, darr)
    losses = [[], []]
    for step in range(6):
        nr, nf = int(rng.integers(600, 1000)), int(rng.integers(600, 1000))
        mk = lambda n: Pairs(torch.from_numpy(rng.integers(0, I, n).astype(np.int32)).to(dev), torch.from_numpy(rng.integers(0, I, n).astype(np.int32)).to(dev))
        real, fake = mk(nr), mk(nf)
        for k, e in enumerate((a, b)):
            losses[k].append(e.d_step(real, fake, 0.7, rng_step=40 + step).clone())
    torch.cuda.synchronize()
    assert a._dfork is not None and b._dfork is None
    if not a._dfork.ok:
        pytest.skip("the aux stream shares a hardware queue with the caller's stream on this box: the step stays on one stream")
    assert a._dfork.expired_waits() == 0 and a._dfork.seq == 6
    for x, y in zip(*losses):
        assert torch.equal(x, y)
    for i in range(8):
        assert torch.equal(a.d_p[i], b.d_p[i]) and torch.equal(a.d_m[i], b.d_m[i]) and torch.equal(a.d_v[i], b.d_v[i]), i
    # a poisoned fork: the Adam sweep returns at once, the host raises when it next looks
    before = [t.clone() for t in a.d_p]
    a._dfork.sync[2] = 1
    a.d_step(real, fake, 0.7, rng_step=99)
    torch.cuda.synchronize()
    for x, y in zip(before, a.d_p):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError, match="gave up"):
        a.check_pipes()


@pytest.mark.parametrize("hs,nr,nf,cuts,dq", [((100, 150, 250, 300), 900, 950, (0, 463, 925, 1388, 1850), "fp32"), ((12, 20, 28, 16), 33, 7, (0, 1, 1, 35, 40), "fp32"),
                                              ((2048, 1024, 512, 256), 260, 250, (0, 255, 510), "fp32"), ((2048, 1024, 512, 256), 260, 250, (0, 255, 510), "fp8"),
                                              ((512, 256, 256, 128), 700, 650, (0, 600, 1350), "fp8")])
def test_d_step_cut_at_the_gradient_exchange_equals_the_step(hs, nr, nf, cuts, dq):
    """ltg_d_grad over disjoint row ranges of the real | fake pair batch (what each rank of a pair-split run computes: the
    ranges straddle the real / fake boundary, one is empty), the gradient vectors summed (the all-reduce), ltg_d_apply ==
    ltg_d_step on the whole batch: same d_loss, same weights and Adam moments (up to the order of the partial sums)."""
    import torch
    from ltgan.engine import Pairs
    I = 500
    rng = np.random.default_rng(5)
    D = O.init_discriminator(I, *hs, seed=3)
    emb, darr = Hh.disc_to_engine(D)
    a, b = _engine(I, "fp32", hs=hs, lr=1e-3, d_precision=dq), _engine(I, "fp32", hs=hs, lr=1e-3, d_precision=dq)
    for e in (a, b):
        e.set_discriminator(emb, darr)
        e.adam_t = 6
    dev = a.device

    def mk(n):
        pop = rng.integers(0, I, n).astype(np.int32)
        nic = rng.integers(0, I, n).astype(np.int32)
        hole = rng.random(n) < 0.05
        pop[hole] = -1
        nic[hole] = -1
        return Pairs(torch.from_numpy(pop).to(dev), torch.from_numpy(nic).to(dev))

    real, fake = mk(nr), mk(nf)
    la = float(a.d_step(real, fake, 0.7, rng_step=31)[0].item())
    total = torch.zeros(b.d_grad_floats(), dtype=torch.float32, device=dev)
    part = torch.empty_like(total)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        b.d_grad(real, fake, lo, hi, part, keep_prob=0.7, rng_step=31)
        total += part
    lb = float(b.d_apply(total)[0].item())
    torch.cuda.synchronize()
    assert a.adam_t == b.adam_t == 7 and abs(la - lb) < 2e-6 * abs(la)
    for i in range(8):
        for xa, xb, what in ((a.d_m[i], b.d_m[i], "m"), (a.d_v[i], b.d_v[i], "v")):
            assert Hh.rel_err(xb.cpu().numpy(), xa.cpu().numpy()) < 2e-5, (what, i)
        diff = (a.d_p[i] - b.d_p[i]).abs()
        if dq == "fp8":
            # quantised operands leave many gradient elements at (or within rounding of) exactly zero; there the first Adam move
            # is lr_t * sign-like(g) and the order of the partial sums decides it: compare where the gradient is a gradient
            diff = diff[a.d_m[i].abs() > 1e-4 * a.d_m[i].abs().max()]
        assert diff.numel() == 0 or diff.max().item() < 0.02 * 1e-3, ("theta", i)     # 2 % of one Adam move


# ------------------------------------------------------------------------------------------------
# sampler (sample.py:40-67 + train.py:227-251) and ranking metrics (eval_functions.py)
# ------------------------------------------------------------------------------------------------
def _sampler_problem(rng, B, I, with_zero_probs):
    cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample = [0], [], [0], [], []
    for b in range(B):
        if b % 7 == 3:                      # invalid user (Q8): no lists, nothing to sample
            n_sample.append(0)
        else:
            n = int(rng.integers(1, 12))
            nc = n + max(2 * n, 10 - n)
            cand_idx += np.sort(rng.choice(I, nc, replace=False)).tolist()
            pop_idx += rng.choice(I, int(rng.integers(1, 9)), replace=False).tolist()
            n_sample.append(n)
        cand_ptr.append(len(cand_idx))
        pop_ptr.append(len(pop_idx))
    valid = (rng.random(I) > 0.03).astype(np.uint8)
    return (np.array(cand_ptr, np.int32), np.array(cand_idx, np.int32), np.array(pop_ptr, np.int32), np.array(pop_idx, np.int32),
            np.array(n_sample, np.int32), valid)


@pytest.mark.parametrize("with_zero_probs", [False, True])
def test_sampler_matches_oracle(with_zero_probs):
    I, B = 1000, 100
    rng = np.random.default_rng(5 + with_zero_probs)
    _sampler_case(rng, I, B, _sampler_problem(rng, B, I, with_zero_probs), with_zero_probs)


def test_sampler_matches_oracle_at_200000_items():
    """BASELINE config 4's item count with the candidate statistics of the C4-shaped synthetic (ltgan.synthetic: a user's n niche
    items + max(2n, 10 - n) others drawn by popularity, up to ~2 500 candidates for the heaviest users)."""
    from ltgan.synthetic import synthetic_index
    idx, _ = synthetic_index("c4", users=100, seed=31)
    rng = np.random.default_rng(41)
    prob = (idx.cand_ptr.astype(np.int32), idx.cand_idx.astype(np.int32), idx.pop_ptr.astype(np.int32), idx.pop_idx.astype(np.int32),
            idx.n_sample.astype(np.int32), idx.valid_item.astype(np.uint8))
    assert int(np.diff(prob[0]).max()) > 20
    _sampler_case(rng, idx.n_items, 100, prob, False)


def test_sampler_matches_oracle_at_20000_items():
    """BASELINE config 3's item count with the candidate statistics of the ML-20M-shaped synthetic (ltgan.synthetic)."""
    from ltgan.synthetic import synthetic_index
    idx, _ = synthetic_index("ml20m", users=100, seed=33)
    rng = np.random.default_rng(43)
    prob = (idx.cand_ptr.astype(np.int32), idx.cand_idx.astype(np.int32), idx.pop_ptr.astype(np.int32), idx.pop_idx.astype(np.int32),
            idx.n_sample.astype(np.int32), idx.valid_item.astype(np.uint8))
    assert idx.n_items == 20000 and int(np.diff(prob[0]).max()) > 20
    _sampler_case(rng, idx.n_items, 100, prob, False)


def _sampler_case(rng, I, B, prob, with_zero_probs):
    import ctypes as C
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    eng = _engine(I, "fp32")
    cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample, valid = prob
    slot_ptr = np.concatenate([[0], np.cumsum(n_sample)]).astype(np.int32)
    acts = eng.new_acts(B)
    logits = rng.normal(0, 2.0, (B, I)).astype(np.float32)
    if with_zero_probs:                     # push most candidates of some users to softmax underflow
        for b in range(0, B, 5):
            c = cand_idx[cand_ptr[b]:cand_ptr[b + 1]]
            if len(c):
                logits[b, c[1:]] = -300.0
    mx = logits.max(1, keepdims=True).astype(np.float64)
    lse = (mx + np.log(np.exp(logits - mx).sum(1, keepdims=True)))[:, 0]
    acts.logits.copy_(torch.from_numpy(logits))
    acts.lse.copy_(torch.from_numpy(lse.astype(np.float32)))
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    d = [t(x) for x in (cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample, slot_ptr, valid)]
    step = 77
    samp = cabi.ltg_sample_inputs(B, int(np.diff(cand_ptr).max()), *[_ptr(x) for x in d], step, None, None, None)
    ns = int(slot_ptr[-1])
    gen = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    pop = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.sample_pairs(samp, acts, gen, pop, cnt)
    torch.cuda.synchronize()
    gen, pop, cnt = gen.cpu().numpy(), pop.cpu().numpy(), int(cnt.cpu().numpy()[0])
    lse32 = lse.astype(np.float32)
    total_ok, flips = 0, 0
    for b in range(B):
        s0, s1 = slot_ptr[b], slot_ptr[b + 1]
        if n_sample[b] == 0:
            continue
        c = cand_idx[cand_ptr[b]:cand_ptr[b + 1]]
        lp = (logits[b, c] - lse32[b]).astype(np.float32)
        p = np.exp(lp.astype(np.float64))
        p[np.exp(lp).astype(np.float32) == 0] = 0.0
        u = O.rng_uniform(SEED, O.STREAM_GUMBEL, step, np.uint64(b) * np.uint64(I) + c.astype(np.uint64))
        want = O.sample_user(c, p, int(n_sample[b]), u)
        k_eff = len(want)
        got_slots = gen[s0:s0 + k_eff]
        assert np.all(gen[s0 + k_eff:s1] == -1) and np.all(pop[s0 + k_eff:s1] == -1)     # unused slots are holes
        up = O.rng_uniform(SEED, O.STREAM_POP_PICK, step, np.uint64(b) * np.uint64(I) + want.astype(np.uint64))
        pl = pop_idx[pop_ptr[b]:pop_ptr[b + 1]]
        xg, xp, kept = O.build_fake_pairs(want, pl, up.astype(np.float32), valid)
        exp_gen = np.where(kept, want, -1)
        if not np.array_equal(got_slots, exp_gen):
            # a flip is only legal where the k-th and (k+1)-th Gumbel keys are within fp32 rounding
            key = np.sort(O.gumbel_keys(p, u))[::-1]
            assert k_eff < len(key) and abs(key[k_eff - 1] - key[k_eff]) < 1e-4, b
            flips += 1
            continue
        exp_pop = np.full(k_eff, -1)
        exp_pop[np.array(kept, bool)] = xp
        assert np.array_equal(pop[s0:s0 + k_eff], exp_pop), b
        total_ok += int(np.sum(kept))
    assert flips <= 1
    if flips == 0:
        assert cnt == total_ok


@pytest.mark.parametrize("n,I", [(64, 1000), (33, 257), (5, 4097), (8, 20000), (4, 200000)])     # the last two: BASELINE configs 3 / 4
def test_rank_metrics_match_oracle(n, I):
    import torch
    from ltgan.engine import CsrRows
    import scipy.sparse as sp
    rng = np.random.default_rng(n + I)
    eng = _engine(I, "fp32")
    pred = rng.random((n, I)).astype(np.float32)
    for c in range(0, I - 1, 17):
        pred[:, c] = pred[:, c + 1]                                    # exact ties
    held = (rng.random((n, I)) < 0.01)
    held[3] = False                                                    # a user without held-out items is dropped
    tr = rng.random((n, I)) < 0.05
    tr[held] = False
    tr[1, :] = True; tr[1, :40] = False; held[1] = False; held[1, :5] = True   # fewer than 100 finite scores
    acts = eng.new_acts(n)
    acts.logits.copy_(torch.from_numpy(pred))
    dev = eng.device
    def rows(m):
        c = sp.csr_matrix(m.astype(np.float32)); c.sort_indices()
        return CsrRows(torch.from_numpy(c.indptr.astype(np.int32)).to(dev), torch.from_numpy(c.indices.astype(np.int32)).to(dev), 0, n)
    out = torch.zeros(n, 4, dtype=torch.float32, device=dev)
    eng.rank_metrics(acts, rows(tr), rows(held), out, 100, 20, 50)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    masked = pred.copy()
    masked[tr] = -np.inf
    nd = O.ndcg_binary_at_k(masked, held, 100)
    r20 = O.recall_at_k(masked, held, 20)
    r50 = O.recall_at_k(masked, held, 50)
    ok = out[:, 3] > 0
    assert ok.sum() == len(nd) and not ok[3]
    np.testing.assert_allclose(out[ok, 0], nd, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out[ok, 1], r20, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out[ok, 2], r50, rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# injected random tensors (ltg.h: "every random tensor can be injected through an optional pointer")
# ------------------------------------------------------------------------------------------------
def test_injected_randomness_g_and_d_steps():
    """keep flags for the input dropout (per CSR entry), eps, the three discriminator masks: fed explicitly, the
    on-device RNG must not be consulted (masks are drawn from numpy here, unrelated to the counter RNG)."""
    import torch
    from ltgan.engine import Pairs
    I, B = 640, 48
    rng, X, P = _problem(I, B, seed=99)
    hs = (20, 24, 40, 36)
    D = O.init_discriminator(I, *hs, seed=6)
    eng = _engine(I, "fp32", hs=hs, lr=1e-3)
    eng.set_generator(Hh.gen_to_engine(P))
    emb, darr = Hh.disc_to_engine(D)
    eng.set_discriminator(emb, darr)
    dev = eng.device
    rows, gen, pop = _fake_pairs(rng, X, I)
    valid = (gen >= 0) & (pop >= 0)
    nf = len(rows)
    keep, dkeep = 0.75, 0.7
    # --- injected tensors
    mask = (rng.random((B, I)) < keep)
    keep_entries = np.concatenate([mask[b, X.indices[X.indptr[b]:X.indptr[b + 1]]] for b in range(B)]).astype(np.uint8)
    eps = rng.standard_normal((B, eng.Z)).astype(np.float32)
    dmf = [(rng.random((nf, w)) < dkeep).astype(np.uint8) for w in hs[1:]]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fake = Pairs(t(pop), t(gen), t(rows))
    batch = _upload_batch(eng, X, with_csc=True)
    acts = eng.new_acts(B)
    cnt = int(valid.sum())
    loss = eng.g_step(batch, fake, acts, torch.tensor([cnt], dtype=torch.int32, device=dev), 0.07, 1.0, keep, 1.0, dkeep, rng_step=5,
                      d_rng_step=6, drop_keep=t(keep_entries), eps=t(eps), drop_fake=[t(m) for m in dmf]).cpu().numpy()
    T = O.d_tower(D, np.where(valid, pop, 0), np.where(valid, gen, 0), [m.astype(np.float64) for m in dmf], dkeep)
    sum_y = float((T["y"] * valid).sum())
    losses, g, F = O.g_loss_and_grads(P, X.toarray(), mask.astype(np.float64), keep, eps.astype(np.float64), 0.07, 1.0, rows[valid],
                                      gen[valid], cnt, sum_y, 1.0, np.float64, quant=False)
    assert abs(loss[4] - sum_y) < 1e-4 * max(1.0, abs(sum_y))
    for i, k in enumerate(("g_loss", "vae_loss", "gan_loss")):
        assert abs(loss[i] - losses[k]) < 1e-3 * abs(losses[k]) + 1e-6, k
    ad = O.SharedAdam(1e-3)
    P64 = {k: np.asarray(v, np.float64) for k, v in P.items()}
    ad.apply(P64, g, O.G_KEYS)
    want_m = Hh.gen_to_engine({k: ad.m[k] for k in O.G_KEYS})
    for i in range(8):
        assert Hh.rel_err(eng.g_m[i].cpu().numpy(), want_m[i]) < 5e-4, ("m", i)
    # --- discriminator step with injected masks for both towers
    nr = 40
    rp = rng.integers(0, I, nr).astype(np.int32)
    rn = rng.integers(0, I, nr).astype(np.int32)
    dmr = [(rng.random((nr, w)) < dkeep).astype(np.uint8) for w in hs[1:]]
    real = Pairs(t(rp), t(rn))
    eng.adam_t = 0
    dl = float(eng.d_step(real, fake, dkeep, rng_step=9, drop_real=[t(m) for m in dmr], drop_fake=[t(m) for m in dmf]).cpu().numpy()[0])
    Tr = O.d_tower(D, rp, rn, [m.astype(np.float64) for m in dmr], dkeep)
    want = -np.log(Tr["y"]).sum() - (np.log(1 - T["y"]) * valid).sum()
    assert abs(dl - want) < 1e-4 * max(1.0, abs(want))
    gr = O.d_tower_backward(D, Tr, [m.astype(np.float64) for m in dmr], dkeep, -(1 - Tr["y"]))
    gf = O.d_tower_backward(D, T, [m.astype(np.float64) for m in dmf], dkeep, T["y"] * valid)
    ad2 = O.SharedAdam(1e-3)
    D64 = {k: np.asarray(v, np.float64) for k, v in D.items()}
    ad2.apply(D64, {k: gr[k] + gf[k] for k in gr}, O.D_KEYS)
    for i, k in enumerate(O.D_KEYS):
        assert Hh.rel_err(eng.d_m[i].cpu().numpy().reshape(-1), ad2.m[k].reshape(-1)) < 5e-4, ("dm", k)


def test_injected_sampler_uniforms():
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    I, B = 500, 40
    rng = np.random.default_rng(17)
    eng = _engine(I, "fp32")
    cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample, valid = _sampler_problem(rng, B, I, False)
    slot_ptr = np.concatenate([[0], np.cumsum(n_sample)]).astype(np.int32)
    ns = int(slot_ptr[-1])
    logits = rng.normal(0, 1.5, (B, I)).astype(np.float32)
    mx = logits.max(1, keepdims=True).astype(np.float64)
    lse = (mx + np.log(np.exp(logits - mx).sum(1, keepdims=True)))[:, 0].astype(np.float32)
    u_g = rng.random(len(cand_idx)).astype(np.float32)
    u_p = rng.random(ns).astype(np.float32)
    acts = eng.new_acts(B)
    acts.logits.copy_(torch.from_numpy(logits))
    acts.lse.copy_(torch.from_numpy(lse))
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    d = [t(x) for x in (cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample, slot_ptr, valid)]
    ug, up = t(u_g), t(u_p)
    samp = cabi.ltg_sample_inputs(B, int(np.diff(cand_ptr).max()), *[_ptr(x) for x in d], 3, _ptr(ug), _ptr(up), None)
    gen = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    pop = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.sample_pairs(samp, acts, gen, pop, cnt)
    torch.cuda.synchronize()
    gen, pop = gen.cpu().numpy(), pop.cpu().numpy()
    total = 0
    for b in range(B):
        if n_sample[b] == 0:
            continue
        c = cand_idx[cand_ptr[b]:cand_ptr[b + 1]]
        p = np.exp((logits[b, c] - lse[b]).astype(np.float64))
        want = O.sample_user(c, p, int(n_sample[b]), u_g[cand_ptr[b]:cand_ptr[b + 1]].astype(np.float64))
        s0 = slot_ptr[b]
        xg, xp, kept = O.build_fake_pairs(want, pop_idx[pop_ptr[b]:pop_ptr[b + 1]], u_p[s0:s0 + len(want)], valid)
        key = np.sort(O.gumbel_keys(p, u_g[cand_ptr[b]:cand_ptr[b + 1]].astype(np.float64)))[::-1]
        k = len(want)
        if k < len(key) and abs(key[k - 1] - key[k]) < 1e-4:
            continue                                            # fp32-vs-fp64 tie at the selection boundary
        assert np.array_equal(gen[s0:s0 + k], np.where(kept, want, -1)), b
        exp_pop = np.full(k, -1)
        exp_pop[np.array(kept, bool)] = xp
        assert np.array_equal(pop[s0:s0 + k], exp_pop), b
        total += int(np.sum(kept))
    assert total > 0


def test_rank_metrics_cut_at_exchange_points_equals_fused():
    """ltg_rank_scores / ltg_rank_counts / ltg_rank_finish over two item slabs (sums done by hand instead of the
    all-reduce) == ltg_rank_metrics on the full rows, bit for bit; ties and masked held-out items included."""
    import torch
    import scipy.sparse as sp
    from ltgan.engine import CsrRows
    I, B = 1000, 37
    rng = np.random.default_rng(23)
    logits = np.round(rng.normal(0, 1, (B, I)), 1).astype(np.float32)           # many exact ties
    X = Hh.random_history(rng, B, I, mean_nnz=30)
    te_r = np.repeat(np.arange(B), 9)
    te = sp.csr_matrix((np.ones(len(te_r), np.float32), (te_r, rng.integers(0, I, len(te_r)))), shape=(B, I))
    te.data[:] = 1.0
    te[5, X[5].indices[:3]] = 1.0                                               # held-out items that are also fold-in items
    te = te.tolil(); te[7] = 0; te = te.tocsr(); te.eliminate_zeros(); te.sort_indices()
    full = _engine(I, "fp32")
    dev = full.device
    t = lambda a, dt=np.int32: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dt))).to(dev)
    te_c = CsrRows(t(te.indptr), t(te.indices), 0, B)
    acts = full.new_acts(B)
    acts.logits.copy_(torch.from_numpy(logits))
    out_full = torch.zeros(B, 4, device=dev)
    full.rank_metrics(acts, CsrRows(t(X.indptr), t(X.indices), 0, B), te_c, out_full)
    cut = 448
    score = torch.zeros(te.nnz, device=dev)
    count = torch.zeros(te.nnz, dtype=torch.int32, device=dev)
    shards = []
    for lo, hi in ((0, cut), (cut, I)):
        e = _engine(I, "fp32", item_lo=lo, item_hi=hi)
        a = e.new_acts(B)
        a.logits.copy_(torch.from_numpy(np.ascontiguousarray(logits[:, lo:hi])))
        Xs = X[:, lo:hi].tocsr(); Xs.sort_indices()
        trc = CsrRows(t(Xs.indptr), t(Xs.indices), 0, B)
        s = torch.zeros(te.nnz, device=dev)
        e.rank_scores(a, trc, te_c, s)
        score += s
        shards.append((e, a, trc))
    for e, a, trc in shards:
        c = torch.zeros(te.nnz, dtype=torch.int32, device=dev)
        e.rank_counts(a, trc, te_c, score, c)
        count += c
    out_cut = torch.zeros(B, 4, device=dev)
    shards[0][0].rank_finish(te_c, count, out_cut)
    torch.cuda.synchronize()
    assert torch.equal(out_full, out_cut)
    # and the oracle on the same matrix
    pred = logits.astype(np.float64).copy()
    pred[X.nonzero()] = -np.inf
    want = O.ndcg_binary_at_k(pred, te.toarray(), 100)
    got = out_cut.cpu().numpy()
    ok = got[:, 3] > 0
    assert ok.sum() == B - 1 and np.allclose(got[ok, 0], want, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# discriminator GEMM precision modes (ltg_config.d_precision; BASELINE config 5 asks for fp8 on the wide discriminator)
# ------------------------------------------------------------------------------------------------
def test_bf16_split_is_exact():
    """d_arith = bf16x6 rests on ONE claim: every fp32 operand x is the exact sum of three bf16 terms, hi = bf16(x), mid = bf16(x - hi),
    lo = bf16(x - hi - mid) (round to nearest even), so that the six cross terms leave out only 2^-26 of a product.  ltg_debug_split runs the
    loaders' own split (csrc/ltg_rgemm.h: ltg_split_bf16) over random values of every magnitude the path meets, powers of two, values on bf16
    rounding boundaries, zeros and signed tiny values; the oracle's bf16 rounding model gives the expected terms bit for bit."""
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    lib = cabi.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.normal(0, 1, 400000), rng.normal(0, 1e-4, 100000), rng.normal(0, 300, 100000), rng.uniform(-1, 1, 100000) * 10.0 ** rng.integers(-20, 20, 100000),
        2.0 ** rng.integers(-60, 60, 1000) * rng.choice([-1, 1], 1000), [0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 + 2.0 ** -8 + 2.0 ** -23, 3.0 * 2.0 ** -9, 65504.0],
        (1.0 + 2.0 ** -8 * (rng.integers(0, 256, 2000) + 0.5)) * 2.0 ** rng.integers(-10, 10, 2000),      # exactly half-way between two bf16 values
    ]).astype(np.float32)
    x = np.concatenate([x, np.zeros((-len(x)) % 4, np.float32)])
    dev = "cuda:0"
    xi = torch.from_numpy(x).to(dev)
    out = torch.empty(3 * len(x), dtype=torch.float32, device=dev)
    cabi.check(lib.ltg_debug_split(_ptr(xi), _ptr(out), len(x), None), "ltg_debug_split")
    torch.cuda.synchronize()
    t = out.cpu().numpy().reshape(-1, 3)
    hi, mid, lo = t[:, 0], t[:, 1], t[:, 2]
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64)), "hi + mid + lo != x"
    want_hi = O.bf16_round(x)
    r1 = (x.astype(np.float64) - want_hi.astype(np.float64)).astype(np.float32)          # exact in fp32
    want_mid = O.bf16_round(r1)
    r2 = (r1.astype(np.float64) - want_mid.astype(np.float64)).astype(np.float32)
    want_lo = O.bf16_round(r2)
    assert np.array_equal(hi.view(np.uint32), want_hi.view(np.uint32)) and np.array_equal(mid.view(np.uint32), want_mid.view(np.uint32))
    assert np.array_equal(lo.view(np.uint32), want_lo.view(np.uint32)) and np.array_equal(lo.astype(np.float64), r2.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= 2.0 ** -8 * np.abs(x[nz])) and np.all(np.abs(lo[nz]) <= 2.0 ** -16 * np.abs(x[nz]))


def test_fp8_rounding_model_matches_hardware():
    """the oracle's e4m3 model (oracle.fp8_e4m3_round) against the conversion the kernels use, bit for bit"""
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    lib = cabi.load()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.normal(0, 1, 20000) * np.exp2(rng.integers(-12, 9, 20000)),
                        np.exp2(np.arange(-14, 9.0)), -np.exp2(np.arange(-14, 9.0)), 1.5 * np.exp2(np.arange(-12, 8.0)),
                        np.arange(-470, 471, 1.0), np.arange(0, 64) * 2.0 ** -10, [0.0, 447.9, 448.0, 463.9, 464.1, 1e6, -1e6]]).astype(np.float32)
    xi = torch.from_numpy(x).cuda()
    out = torch.empty_like(xi)
    cabi.check(lib.ltg_fp8_roundtrip(_ptr(xi), _ptr(out), x.size, None), "ltg_fp8_roundtrip")
    torch.cuda.synchronize()
    want = O.fp8_e4m3_round(x.astype(np.float64)).astype(np.float32)
    got = out.cpu().numpy()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (x[bad[:8]], got[bad[:8]], want[bad[:8]])


# fp8 tolerances = the pipeline's own noise floor: the fp8 MFMA accumulates with ~1.5e-5 relative error (ltg_debug_gemm,
# test below); feeding the ORACLE operands perturbed by that much moves its w1 gradient by 0.9 % (default sizes) / 3.2 %
# (wide) in the energy norm, because every value near an e4m3 rounding boundary flips by a 6 % step.
@pytest.mark.parametrize("dq,hs,tol", [("bf16", (100, 150, 250, 300), 2e-3), ("fp8", (100, 150, 250, 300), 2e-2),
                                       ("bf16", (2048, 1024, 512, 256), 2e-3), ("fp8", (2048, 1024, 512, 256), 6e-2),
                                       ("fp32", (2048, 1024, 512, 256), 5e-4)])
def test_d_step_precision_modes(dq, hs, tol):
    """D step with quantised GEMM operands == oracle fed the SAME quantised operands (d6: compare against the oracle
    fed rounded operands); and close to the fp32 discriminator within the format's own error."""
    import torch
    from ltgan.engine import Pairs
    I, nr, nf = 700, 300, 280
    rng = np.random.default_rng(41)
    D = O.init_discriminator(I, *hs, seed=5)
    for k in ("b1", "b2", "b3", "b4"):
        D[k] = rng.normal(0, 0.05, D[k].shape).astype(np.float32)
    eng = _engine(I, "fp32", hs=hs, lr=1e-3, d_precision=dq)
    emb, darr = Hh.disc_to_engine(D)
    eng.set_discriminator(emb, darr)
    dev = eng.device
    rp, rn = rng.integers(0, I, nr).astype(np.int32), rng.integers(0, I, nr).astype(np.int32)
    fp, fn = rng.integers(0, I, nf).astype(np.int32), rng.integers(0, I, nf).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    step, keep = 4, 0.7
    loss = float(eng.d_step(Pairs(t(rp), t(rn)), Pairs(t(fp), t(fn)), keep, rng_step=step).cpu().numpy()[0])
    dm = Hh.d_masks(SEED, step, nr + nf, hs[1:], keep)
    mode = None if dq == "fp32" else dq
    Tr = O.d_tower(D, rp, rn, [m[:nr] for m in dm], keep, dq=mode)
    Tf = O.d_tower(D, fp, fn, [m[nr:] for m in dm], keep, dq=mode)
    want_loss = -np.log(Tr["y"]).sum() - np.log(1 - Tf["y"]).sum()
    assert abs(loss - want_loss) < 1e-4 * abs(want_loss), (loss, want_loss)
    gr = O.d_tower_backward(D, Tr, [m[:nr] for m in dm], keep, -(1 - Tr["y"]), dq=mode)
    gf = O.d_tower_backward(D, Tf, [m[nr:] for m in dm], keep, Tf["y"], dq=mode)
    ad = O.SharedAdam(1e-3)
    D64 = {k: np.asarray(v, np.float64) for k, v in D.items()}
    ad.apply(D64, {k: gr[k] + gf[k] for k in gr}, O.D_KEYS)
    for i, k in enumerate(O.D_KEYS):
        got, want = eng.d_m[i].cpu().numpy().reshape(-1).astype(np.float64), ad.m[k].reshape(-1)
        # energy norm: a value that sits on a rounding boundary of the operand format may round the other way on the GPU
        # (fp32 tanh / products vs the fp64 oracle); one flipped e4m3 operand moves one product term by 6 %
        assert np.linalg.norm(got - want) < tol * np.linalg.norm(want), ("m", k)
        assert Hh.rel_err(got, want) < tol, ("m max", k)
    if dq == "fp8":
        # The oracle above rounds d a / d pre of the branch layers to bf16 BECAUSE the device stores it so (dA1T_16): for that term the comparison
        # is against a model of the device.  The cost of the approximation is pinned here against the pipeline WITHOUT the storage rounding: the
        # only tensors behind it are w1 / b1 / w2 / b2, whose gradient operand dpre1 = dhin * (d a / d pre) moves by at most the bf16 half-ulp
        # 2^-9 relative per element BEFORE it is quantised to e4m3 -- whose own step is 2^-4, so an element near a rounding boundary flips by 6 %:
        # measured oracle against oracle 1.1-1.5 % of the first moments in the energy norm at both sizes; bound 2 %.
        gr0 = O.d_tower_backward(D, Tr, [m[:nr] for m in dm], keep, -(1 - Tr["y"]), dq=mode, dact16=False)
        gf0 = O.d_tower_backward(D, Tf, [m[nr:] for m in dm], keep, Tf["y"], dq=mode, dact16=False)
        ad0 = O.SharedAdam(1e-3)
        ad0.apply({k: np.asarray(v, np.float64) for k, v in D.items()}, {k: gr0[k] + gf0[k] for k in gr0}, O.D_KEYS)
        for i, k in enumerate(O.D_KEYS):
            got, want = eng.d_m[i].cpu().numpy().reshape(-1).astype(np.float64), ad0.m[k].reshape(-1)
            assert np.linalg.norm(got - want) < (tol + 2e-2) * np.linalg.norm(want), ("m vs the un-rounded d a / d pre", k)
            # and the two oracle variants themselves: the rounding alone, no device in it
            assert np.linalg.norm(ad.m[k].reshape(-1) - want) < 2e-2 * np.linalg.norm(want), ("bf16 storage of d a / d pre", k)
    if mode is not None:       # how far the format itself moves the result from the fp32 discriminator
        T32 = O.d_tower(D, rp, rn, [m[:nr] for m in dm], keep)
        lim = 2e-2 if dq == "bf16" else 0.25
        assert np.max(np.abs(Tr["y"] - T32["y"])) < lim


def test_fp8_operand_shadows_stay_in_step_with_the_master_weights():
    """LTG_PREC_FP8 with operand-format storage (wide discriminator): after several D steps the transposed e4m3 weight
    shadows the Adam sweep maintains equal a fresh rebuild from the fp32 master weights, bit for bit; and the forward fed
    from them matches the on-the-fly conversion path (tuning-knob bit 18) to fp32 round-off of the accumulation order."""
    import ctypes as C
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import Pairs
    I, hs = 600, (2048, 1024, 512, 256)
    rng = np.random.default_rng(9)
    a, b = _engine(I, "fp32", hs=hs, lr=1e-3, d_precision="fp8"), _engine(I, "fp32", hs=hs, lr=1e-3, d_precision="fp8")
    assert a.d_fp8 is not None
    b.cfg.tuning = 262144                      # round-1 path: operands converted on the fly from fp32
    b.set_discriminator(a.d_emb.cpu().numpy(), [p.cpu().numpy() for p in a.d_p])
    dev = a.device
    t = lambda x: torch.from_numpy(x.astype(np.int32)).to(dev)
    real = Pairs(t(rng.integers(0, I, 200)), t(rng.integers(0, I, 200)))
    fake = Pairs(t(rng.integers(0, I, 190)), t(rng.integers(0, I, 190)))
    for k in range(3):
        la = float(a.d_step(real, fake, 0.7, rng_step=5 + k)[0].item())
        lb = float(b.d_step(real, fake, 0.7, rng_step=5 + k)[0].item())
        assert abs(la - lb) < 2e-3 * abs(lb), (k, la, lb)      # fp8 pipeline: a flipped rounding moves one operand by 6 %
    # ... and after a step cut at its gradient exchange (ltg_d_grad -> ltg_d_apply: the pair-split path of a sharded run)
    g = torch.empty(a.d_grad_floats(), dtype=torch.float32, device=dev)
    a.d_grad(real, fake, 0, 390, g, keep_prob=0.7, rng_step=9)
    a.d_apply(g)
    kept = [x.clone() for x in a.d_fp8]
    cabi.check(a.lib.ltg_refresh_d_shadow(C.byref(a.cfg), C.byref(a.disc_c), a.stream()), "ltg_refresh_d_shadow")
    torch.cuda.synchronize()
    assert len(kept) == 5
    for name, x, y in zip(("emb", "w1t", "w2t", "w3t", "w3"), kept, a.d_fp8):
        assert torch.equal(x, y), name
    assert int((kept[1] != 0).sum()) > 0.9 * kept[1].numel() and int((kept[4] != 0).sum()) > 0.9 * kept[4].numel()


@pytest.mark.parametrize("hs,nr,nf", [((2048, 1024, 512, 256), 900, 921), ((512, 256, 256, 128), 130, 61), ((256, 128, 128, 128), 1, 0)])
def test_fp8_backward_in_operand_format_equals_the_on_the_fly_conversion(hs, nr, nf):
    """BASELINE config 5: with the fifth shadow (w3 in its own layout) every BACKWARD product of the discriminator step reads e4m3
    bytes in operand format too (csrc/ltg_fp8bwd.h: transposed copies of A1, dpre3, dpre1 and of the gathered embedding rows, pair
    rows zero-padded to a multiple of 128, 1024-row split-K chunks).  Same static scales and conversion as the path that converts
    fp32 operands inside the GEMM loaders (tuning-knob bit 19), so both feed the MFMAs identical operand values: d_loss and every
    Adam moment agree to the round-off of the fp32 accumulation order -- whole step, and cut at its gradient exchange."""
    import torch
    from ltgan.engine import Pairs
    I = 500
    rng = np.random.default_rng(77)
    a, b = _engine(I, "fp32", hs=hs, lr=1e-3, d_precision="fp8"), _engine(I, "fp32", hs=hs, lr=1e-3, d_precision="fp8")
    b.cfg.tuning = 1 << 19                     # forward from the shadows, backward converts on the fly (the round-2 path)
    b.set_discriminator(a.d_emb.cpu().numpy(), [p.cpu().numpy() for p in a.d_p])
    dev = a.device
    t = lambda x: torch.from_numpy(x.astype(np.int32)).to(dev)
    rp, rn, fp, fn = (rng.integers(0, I, k) for k in (nr, nr, nf, nf))
    if nf > 3:
        fp[2] = -1                                # holes (dropped pairs, Q9 / Q10)
        fn[2] = -1
    real, fake = Pairs(t(rp), t(rn)), Pairs(t(fp), t(fn))
    for k in range(2):
        la = float(a.d_step(real, fake, 0.7, rng_step=5 + k)[0].item())
        lb = float(b.d_step(real, fake, 0.7, rng_step=5 + k)[0].item())
        assert abs(la - lb) <= 1e-5 * abs(lb) + 1e-6, (k, la, lb)
        for i in range(8):
            x, y = a.d_m[i].double(), b.d_m[i].double()
            assert (x - y).norm().item() <= 2e-4 * y.norm().item() + 1e-12, ("m", k, i)
            if k == 0:                            # one step from equal weights: v = (1 - b2) g^2 compares the squared gradients
                assert (a.d_v[i].double() - b.d_v[i].double()).norm().item() <= 4e-4 * b.d_v[i].double().norm().item() + 1e-20, ("v", i)
    # ... and cut at the gradient exchange (ltg_d_grad over a row range), from IDENTICAL weights again (two steps of round-off in the
    # weights flip e4m3 roundings of the operands: an fp8 pipeline amplifies that to the percent level)
    b.set_discriminator(a.d_emb.cpu().numpy(), [p.cpu().numpy() for p in a.d_p])
    n = nr + nf
    ga = torch.empty(a.d_grad_floats(), dtype=torch.float32, device=dev)
    gb = torch.empty_like(ga)
    lo, hi = n // 3, n - n // 4
    a.d_grad(real, fake, lo, hi, ga, keep_prob=0.7, rng_step=11)
    b.d_grad(real, fake, lo, hi, gb, keep_prob=0.7, rng_step=11)
    torch.cuda.synchronize()
    L = a.d_grad_floats()
    P = sum(int(x.numel()) for x in a.d_p)
    err = (ga[:P].double() - gb[:P].double()).norm().item() / gb[:P].double().norm().item()
    assert err <= 1e-3 and abs(float(ga[P]) - float(gb[P])) <= 1e-5 * abs(float(gb[P])) + 1e-6, (err, float(ga[P]), float(gb[P]), L, P)


def test_gemm_block_operand_modes():
    """The MFMA block template against exact products of the operands each mode feeds it (ltg_debug_gemm): fp32 and bf16
    accumulate to fp32 round-off; the fp8 MFMA of gfx950 accumulates with ~1.5e-5 relative error (measured, asserted)."""
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    lib = cabi.load()
    rng = np.random.default_rng(0)
    for (M, N, K) in ((64, 64, 100), (100, 96, 400), (33, 70, 2048)):
        A = rng.normal(0, 1, (M, K)).astype(np.float32)
        B = rng.normal(0, 1, (K, N)).astype(np.float32)
        a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        for mode, lim in ((0, 3e-6), (1, 3e-6), (2, 1e-4)):
            c = torch.zeros(M, N, device="cuda")
            cabi.check(lib.ltg_debug_gemm(mode, M, N, K, _ptr(a), _ptr(b), _ptr(c), None), "ltg_debug_gemm")
            torch.cuda.synchronize()
            if mode == 0:
                qa, qb = A.astype(np.float64), B.astype(np.float64)
            elif mode == 1:
                qa, qb = O.bf16_round(A).astype(np.float64), O.bf16_round(B).astype(np.float64)
            else:
                qa, qb = O.fp8_e4m3_round(A.astype(np.float64) * 16) / 16, O.fp8_e4m3_round(B.astype(np.float64) * 16) / 16
            want = qa @ qb
            assert Hh.rel_err(c.cpu().numpy(), want) < lim, (M, N, K, mode)


def test_g_step_without_slot_cache_is_bit_identical():
    """ltg_batch.slot == NULL: the library builds the item -> gradient-row map of the batch in its workspace; same bits."""
    import torch
    from ltgan.engine import CsrRows, Pairs
    I, B = 1200, 64
    rng, X, P = _problem(I, B, seed=77)
    rows, gen, pop = _fake_pairs(rng, X, I)
    outs = []
    for with_slot in (True, False):
        eng = _engine(I, "bf16", lr=1e-3)
        eng.set_generator(Hh.gen_to_engine(P))
        dev = eng.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
        batch = CsrRows(t(X.indptr.astype(np.int32)), t(X.indices.astype(np.int32)), 0, B, slot=t(slot) if with_slot else None, uptr=t(uptr),
                        rowidx=t(rowidx), csr_pos=t(pos), n_unique=nu)
        acts = eng.new_acts(B)
        fake = Pairs(t(pop), t(gen), t(rows))
        cnt = torch.tensor([int(((gen >= 0) & (pop >= 0)).sum())], dtype=torch.int32, device=dev)
        for step in range(3):
            loss = eng.g_step(batch, fake, acts, cnt, 0.05, rng_step=10 + step, d_rng_step=20 + step).clone()
        torch.cuda.synchronize()
        outs.append((loss.cpu(), [p.cpu() for p in eng.g_p], [m.cpu() for m in eng.g_m]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
        assert torch.equal(a, b)


def cabi_flags(*names):
    from ltgan import _cabi as cabi
    v = 0
    for n in names:
        v |= getattr(cabi, n)
    return v


@pytest.mark.parametrize("precision,period", [("fp32", 3), ("bf16", 5)])
def test_lazy_adam_clock_of_the_first_encoder_layer_is_bit_identical_to_the_dense_sweep(precision, period):
    """ltg_gen_state.q0_last: TF's Adam (train.py:160-164) moves every row of W_q0 every step; the lazy clock applies a row's
    zero-gradient steps later, with the same arithmetic.  Several G steps over DIFFERENT batches (rows go in and out of the
    batches, the rotating slice has a short period), a forward over other users in the middle of the phase, one flush at
    the end: every parameter, moment and output bit equals the dense sweep's."""
    import torch
    from ltgan.engine import CsrRows, Pairs
    I, B, n_batches = 9000, 48, 7
    rng = np.random.default_rng(4242)
    P = O.init_generator(I, seed=3)
    Xs = [Hh.random_history(rng, B, I, mean_nnz=14) for _ in range(n_batches)]
    Xf = Hh.random_history(rng, 40, I, mean_nnz=30)                       # forward-only batch (no distinct-item list)
    fakes = [_fake_pairs(rng, X, I) for X in Xs]
    outs = []
    # "one-call": the lazy clock inside ltg_g_step_sharded -- the decoder weight update runs beside the next step's encoder half, the slice
    # of step t on the side stream between the catch-ups of steps t + 1 and t + 2, everything handed over through device words (bf16
    # decoder path only); "-slice-in-touch": the slice rides in the catch-up launch of step t + 1 (a row in both sets goes to whichever
    # workgroup claims it)
    # "-no-uitem": the batch without its list of distinct items (ltg_batch.uitem, ABI v11): the kernels that walk the distinct items then
    # find an item through uptr -> csr_pos -> indices, as before
    # "-wide-grad": the sparse gradient in its column-blocked shape (the default of the one-call step is one wave per row over all columns)
    # "-events": fork / join of the weight update as event pairs instead of device words
    # "-tail-own": the Adam tail on the pipe's third stream beside the sparse gradient kernel and the next call's enc-0, its bias job re-summing
    # the partial bias rows in the gradient kernel's order (the default: the step's last kernel on the caller's stream)
    pipe_flags = {"one-call": 0, "one-call-events": cabi_flags("LTG_PIPE_EVENTS"), "one-call-slice-in-touch": cabi_flags("LTG_PIPE_SLICE_IN_TOUCH"),
                  "one-call-no-uitem": 0, "one-call-tail-own": cabi_flags("LTG_PIPE_TAIL_OWN"),
                  "one-call-wide-grad": cabi_flags("LTG_PIPE_WIDE_GRAD")}
    for variant in ("dense", "lazy", "lazy-no-uitem") + (tuple(pipe_flags) if precision == "bf16" else ()):
        lazy = variant != "dense"
        eng = _engine(I, precision, lr=1e-3, lazy_q0=lazy, q0_period=period)
        assert eng.lazy_q0 == lazy
        eng.set_generator(Hh.gen_to_engine(P))
        dev = eng.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        acts = eng.new_acts(B)
        eng.q0_defer = True
        losses, mids = [], []
        pipe = None
        if variant in pipe_flags:
            from ltgan.engine import Pipe
            assert eng.sharded_step_ok(B)
            pipe = Pipe(eng, B, flags=pipe_flags[variant] | int(os.environ.get("LTGAN_TEST_PIPE_FLAGS", "0")))
        batches = []
        for X in Xs:
            slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
            uitem = None if variant.endswith("no-uitem") else t(np.flatnonzero(slot >= 0).astype(np.int32))
            batches.append(CsrRows(t(X.indptr.astype(np.int32)), t(X.indices.astype(np.int32)), 0, B, uptr=t(uptr), rowidx=t(rowidx), csr_pos=t(pos),
                                   n_unique=nu, uitem=uitem, uitem_off=0))
        for s in range(2 * n_batches):
            batch = batches[s % n_batches]
            rows, gen, pop = fakes[s % n_batches]
            fake = Pairs(t(pop), t(gen), t(rows))
            cnt = torch.tensor([int(((gen >= 0) & (pop >= 0)).sum())], dtype=torch.int32, device=dev)
            eng.adam_t += s % 3                                           # the shared counter also moves between G steps (D steps)
            if pipe is not None:
                go = eng.g_opts(cnt, 0.05, rng_step=10 + s, d_rng_step=200 + s)
                # the NEXT batch announced: its rows of W_q0 are caught up during this call and the next call launches no catch-up (ABI v13) --
                # also across the mid-phase forward below, which moves no ordinal
                losses.append(eng.g_step_sharded(batch, fake, acts, go, pipe, next_batch=batches[(s + 1) % n_batches]).clone())
            else:
                losses.append(eng.g_step(batch, fake, acts, cnt, 0.05, rng_step=10 + s, d_rng_step=200 + s).clone())
            if s == n_batches + 2:                                        # mid-phase forward: rows are caught up as they are read
                if pipe is not None:
                    eng.pipe_join(pipe)                                   # (the forked pieces of the last step first)
                fa = eng.new_acts(40)
                eng.forward(CsrRows(t(Xf.indptr.astype(np.int32)), t(Xf.indices.astype(np.int32)), 0, 40), fa)
                mids.append(fa.logits[:40].clone())
                mids.append(fa.h1[:40].clone())
        if pipe is not None:
            eng.pipe_join(pipe)
            assert pipe.expired_waits() == 0, variant
            ahead_served = variant not in ("one-call-events", "one-call-slice-in-touch", "one-call-no-uitem") and pipe.handover != "events"
            assert pipe.ahead_calls == (2 * n_batches - 1 if ahead_served else 0), (variant, pipe.handover, pipe.ahead_calls)
        if lazy:
            torch.cuda.synchronize()
            assert eng.gen_c.q0_ord == 2 * n_batches
            assert int(eng.q0_last.min()) < eng.gen_c.q0_ord              # some rows really lag before the flush
            # ... by no more than the period (the one-call step runs a step's slice one step later)
            assert int(eng.q0_last.min()) >= eng.gen_c.q0_ord - period - (1 if pipe is not None else 0)
        eng.q0_defer = False
        eng.g_flush()
        torch.cuda.synchronize()
        if lazy:
            assert eng.gen_c.q0_ord == 0 and int(eng.q0_last.max()) == 0    # every row current: the ordinals restart
        outs.append([x.cpu() for x in losses + mids + eng.g_p + eng.g_m + eng.g_v])
    names = ["dense", "lazy", "lazy-no-uitem"] + (list(pipe_flags) if precision == "bf16" else [])
    for name, o in zip(names[1:], outs[1:]):
        assert len(o) == len(outs[0])
        for k, (a, b) in enumerate(zip(outs[0], o)):
            assert torch.equal(a, b), (name, k)


def test_sampler_ties_and_long_candidate_lists():
    """Equal Gumbel keys go to the smaller index (oracle: lexsort on (-key, index)); the kernel's fast pass counts strictly
    larger keys only and must fall back to the exact ranks when equal keys straddle the selection boundary.  Users:
    0 = every key equal, 1 = 2 500 candidates (several passes of the 1 024-thread workgroup), 2 = a block of equal maximal
    keys larger than the sample, 3 = equal keys entirely inside the sample (fast pass suffices)."""
    import torch
    from ltgan import _cabi as cabi
    from ltgan.engine import _ptr
    I, B = 6000, 4
    rng = np.random.default_rng(99)
    eng = _engine(I, "fp32")
    ncs, nsamp = [50, 2500, 300, 300], [7, 40, 9, 30]
    cand = [np.sort(rng.choice(I, n, replace=False)).astype(np.int32) for n in ncs]
    cand_ptr = np.concatenate([[0], np.cumsum(ncs)]).astype(np.int32)
    cand_idx = np.concatenate(cand)
    pop_ptr = np.arange(B + 1, dtype=np.int32) * 3
    pop_idx = rng.choice(I, 3 * B, replace=False).astype(np.int32)
    n_sample = np.array(nsamp, np.int32)
    slot_ptr = np.concatenate([[0], np.cumsum(n_sample)]).astype(np.int32)
    valid = np.ones(I, np.uint8)
    logits = rng.normal(0, 1.5, (B, I)).astype(np.float32)
    u_g = rng.random(len(cand_idx)).astype(np.float32)
    logits[0, cand[0]] = 0.25
    u_g[cand_ptr[0]:cand_ptr[1]] = 0.5
    tie = rng.choice(ncs[2], 40, replace=False)                  # user 2: 40 equal keys above everything else, 9 to sample
    logits[2, cand[2][tie]] = 9.0
    u_g[cand_ptr[2] + tie] = 0.75
    tie3 = rng.choice(ncs[3], 12, replace=False)                 # user 3: 12 equal top keys, 30 to sample
    logits[3, cand[3][tie3]] = 9.0
    u_g[cand_ptr[3] + tie3] = 0.75
    mx = logits.max(1, keepdims=True).astype(np.float64)
    lse = (mx + np.log(np.exp(logits - mx).sum(1, keepdims=True)))[:, 0].astype(np.float32)
    ns = int(slot_ptr[-1])
    u_p = rng.random(ns).astype(np.float32)
    acts = eng.new_acts(B)
    acts.logits[:B].copy_(torch.from_numpy(logits))
    acts.lse[:B].copy_(torch.from_numpy(lse))
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    d = [t(x) for x in (cand_ptr, cand_idx, pop_ptr, pop_idx, n_sample, slot_ptr, valid)]
    ug, up = t(u_g), t(u_p)
    samp = cabi.ltg_sample_inputs(B, max(ncs), *[_ptr(x) for x in d], 3, _ptr(ug), _ptr(up), None)
    gen = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    pop = torch.full((ns,), -7, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.sample_pairs(samp, acts, gen, pop, cnt)
    torch.cuda.synchronize()
    gen = gen.cpu().numpy()
    assert int(cnt.item()) == ns
    for b in range(B):
        c = cand[b]
        lp = (logits[b, c] - lse[b]).astype(np.float32)
        key32 = lp - np.log(-np.log(u_g[cand_ptr[b]:cand_ptr[b + 1]].astype(np.float32))).astype(np.float32)
        want = np.sort(np.lexsort((np.arange(len(c)), -key32.astype(np.float64)))[:nsamp[b]])
        got = gen[slot_ptr[b]:slot_ptr[b + 1]]
        if b in (0, 2, 3):                                       # the ties are exact in any precision: the answer is determined
            assert np.array_equal(got, c[want]), b
        else:                                                    # fp32 keys of the kernel vs numpy's: allow a boundary flip
            assert len(np.setdiff1d(got, c[want])) <= 1, b
    assert np.array_equal(gen[slot_ptr[0]:slot_ptr[1]], cand[0][:7])                       # all equal: the first seven
    assert np.array_equal(gen[slot_ptr[2]:slot_ptr[3]], cand[2][np.sort(tie)[:9]])         # the first nine of the tied block


def test_streaming_decoder_storage_choices_are_bit_identical():
    """Large item slabs, bf16 decoder: (a) dlogits stored as bf16 for the streaming consumers (they round it to bf16 for the
    MFMA anyway) vs fp32 storage (tuning-knob bit 22), (b) softmax statistics from the decoder epilogue vs a second pass over
    the logits (bit 21: same partial format, another summation order -> compared with a tolerance).  I = 8264: 258 full
    32-item tiles + a ragged tail of 8 rows for the generic weight-gradient kernel."""
    import torch
    from ltgan.engine import CsrRows, Pairs
    I, B = 8264, 100
    rng, X, P = _problem(I, B, seed=31)
    rows, gen, pop = _fake_pairs(rng, X, I)
    outs = {}
    for knob in (0, 1 << 22, 1 << 21):
        eng = _engine(I, "bf16", lr=1e-3)
        eng.cfg.tuning = knob
        eng.set_generator(Hh.gen_to_engine(P))
        dev = eng.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        slot, uptr, rowidx, pos, nu = Hh.csc_view(X)
        batch = CsrRows(t(X.indptr.astype(np.int32)), t(X.indices.astype(np.int32)), 0, B, uptr=t(uptr), rowidx=t(rowidx), csr_pos=t(pos), n_unique=nu)
        acts = eng.new_acts(B)
        fake = Pairs(t(pop), t(gen), t(rows))
        cnt = torch.tensor([int(((gen >= 0) & (pop >= 0)).sum())], dtype=torch.int32, device=dev)
        for step in range(3):
            loss = eng.g_step(batch, fake, acts, cnt, 0.05, rng_step=10 + step, d_rng_step=20 + step).clone()
        torch.cuda.synchronize()
        outs[knob] = [loss.cpu()] + [p.cpu() for p in eng.g_p] + [m.cpu() for m in eng.g_m] + [acts.lse[:B].cpu()]
    for a, b in zip(outs[0], outs[1 << 22]):
        assert torch.equal(a, b)
    # another summation order of the row statistics: lse moves in its last bits, a dlogit may round to the neighbouring bf16,
    # and Adam turns any gradient difference into a fraction of lr = 1e-3 per step
    for k, (a, b) in enumerate(zip(outs[0], outs[1 << 21])):
        if k in (0, len(outs[0]) - 1):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), k
        else:       # an element whose gradient is within rounding of zero moves by up to lr in either direction
            assert float((a - b).abs().max()) <= 1e-3 and float((a - b).abs().mean()) <= 2e-6, k
