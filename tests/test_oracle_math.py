"""Pins the oracle's closed-form gradients / Adam against torch-CPU autograd (the reference's
arithmetic lives in TensorFlow, which is not installable here: SURVEY 8/c1-c2)."""
import numpy as np
import torch

import helpers as Hh
from oracle import ltg_oracle as O


def _torch_g_loss(P, X, mask, keep, eps, anneal, lam, S_rows, S_cols, cnt, sum_y):
    t = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in P.items()}
    X = torch.tensor(X, dtype=torch.float64)
    h = torch.nn.functional.normalize(X, dim=1, eps=1e-12)            # tf.nn.l2_normalize
    h = h / keep * torch.tensor(mask)
    h1 = torch.tanh(h @ t["Wq0"] + t["bq0"])
    a2 = h1 @ t["Wq1"] + t["bq1"]
    mu, lv = a2[:, :O.Z_DIM], a2[:, O.Z_DIM:]
    std = torch.exp(0.5 * lv)
    KL = (0.5 * (-lv + torch.exp(lv) + mu ** 2 - 1)).sum(1).mean()
    z = mu + torch.tensor(eps) * std
    h2 = torch.tanh(z @ t["Wp0"] + t["bp0"])
    logits = h2 @ t["Wp1"] + t["bp1"]
    neg_ll = -(torch.log_softmax(logits, 1) * X).sum(1).mean()
    probs = torch.softmax(logits, 1)
    gen_mask = torch.zeros_like(probs)
    gen_mask[S_rows, S_cols] = 1.0
    sampled = (probs * gen_mask).reshape(-1)
    nz = sampled[sampled != 0]                                        # train.py:149
    y = torch.full((len(S_rows), 1), sum_y / max(1, len(S_rows)), dtype=torch.float64)  # any y with that sum (Q2)
    gan = -(lam / cnt) * (nz * y).sum()                               # [K]*[K,1] broadcast, train.py:157
    loss = neg_ll + anneal * KL + gan
    loss.backward()
    return loss.item(), (neg_ll + anneal * KL).item(), gan.item(), {k: v.grad.numpy() for k, v in t.items()}


def test_generator_grads_match_autograd():
    rng = np.random.default_rng(0)
    I, B = 120, 9
    X = Hh.random_history(rng, B, I, mean_nnz=7).toarray()
    P = O.init_generator(I, seed=3)
    mask = (rng.random((B, I)) < 0.75).astype(np.float64)
    eps = rng.standard_normal((B, O.Z_DIM))
    S_rows = np.array([0, 0, 2, 5, 5, 5])
    S_cols = np.array([3, 17, 40, 1, 2, 99])
    losses, g, _ = O.g_loss_and_grads(P, X, mask, 0.75, eps, 0.17, 1.3, S_rows, S_cols, 6, 2.71)
    L, vae, gan, tg = _torch_g_loss(P, X, mask, 0.75, eps, 0.17, 1.3, S_rows, S_cols, 6, 2.71)
    assert abs(losses["g_loss"] - L) < 1e-10
    assert abs(losses["vae_loss"] - vae) < 1e-10
    assert abs(losses["gan_loss"] - gan) < 1e-10
    for k in O.G_KEYS:
        np.testing.assert_allclose(g[k], tg[k].reshape(g[k].shape), rtol=1e-8, atol=1e-12, err_msg=k)


def test_discriminator_grads_match_autograd():
    rng = np.random.default_rng(1)
    I, hs, keep = 50, (8, 6, 10, 7), 0.7
    D = O.init_discriminator(I, *hs, seed=2)
    nr, nf = 13, 11
    real = (rng.integers(0, I, nr), rng.integers(0, I, nr), [(rng.random((nr, w)) < keep).astype(float) for w in hs[1:]])
    fake = (rng.integers(0, I, nf), rng.integers(0, I, nf), [(rng.random((nf, w)) < keep).astype(float) for w in hs[1:]])
    loss, g, _, _ = O.d_loss_and_grads(D, real, fake, keep)
    t = {k: torch.tensor(np.asarray(D[k], np.float64), requires_grad=(k != "emb")) for k in D}

    def tower(pop, nic, m):
        mA, mB, mC = (torch.tensor(x) for x in m)
        ea, eb = t["emb"][torch.tensor(pop)], t["emb"][torch.tensor(nic)]
        a = torch.tanh(ea @ t["w1"] + t["b1"]) / keep * mA
        b = torch.tanh(eb @ t["w2"] + t["b2"]) / keep * mB
        c = torch.tanh(torch.cat([a, b], 1) @ t["w3"] + t["b3"]) / keep * mC
        return torch.sigmoid(c @ t["w4"] + t["b4"])

    L = -torch.log(tower(*real)).sum() - torch.log(1 - tower(*fake)).sum()
    L.backward()
    assert abs(L.item() - loss) < 1e-10
    for k in O.D_KEYS:
        np.testing.assert_allclose(g[k].reshape(-1), t[k].grad.numpy().reshape(-1), rtol=1e-8, atol=1e-12, err_msg=k)


def test_shared_adam_matches_torch_form():
    """theta -= lr_t*m/(sqrt(v)+eps) with ONE step counter across two variable groups (Q5)."""
    ad = O.SharedAdam(1e-3)
    P = {"a": np.ones(3), "b": np.ones(2)}
    ad.apply(P, {"a": np.array([1.0, -2.0, 0.0])}, ["a"])      # t = 1 (a "D" step)
    ad.apply(P, {"b": np.array([0.5, 0.0])}, ["b"])            # t = 2 (a "G" step) -- b's FIRST update uses t=2
    lr2 = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    m, v = 0.1 * 0.5, 0.001 * 0.25
    assert abs(P["b"][0] - (1 - lr2 * m / (np.sqrt(v) + 1e-8))) < 1e-15
    assert P["b"][1] == 1.0 and ad.t == 2
    # zero gradient still decays m and moves theta (dense Adam)
    ad.apply(P, {"a": np.zeros(3)}, ["a"])
    assert abs(ad.m["a"][0] - 0.9 * 0.1) < 1e-15 and P["a"][0] < 1 - 1e-4


def test_rng_is_24bit_and_reproducible():
    u = O.rng_uniform(7, O.STREAM_GUMBEL, 3, np.arange(1000))
    assert np.all(u >= 0) and np.all(u < 1) and np.all(u * 2 ** 24 == np.floor(u * 2 ** 24))
    assert np.array_equal(u, O.rng_uniform(7, O.STREAM_GUMBEL, 3, np.arange(1000)))
    assert abs(u.mean() - 0.5) < 0.03
    n = O.rng_normal(7, O.STREAM_VAE_EPS, 3, np.arange(20000))
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1) < 0.03
    # known answers (pin the hash against accidental edits; the HIP kernels use the same constants)
    assert int(O.rng_u64(1, 1, 0, 0)) == int(O.rng_u64(1, 1, 0, np.array([0]))[0])


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.005859375, -1.005859375, 3.140625], np.float32)
    r = O.bf16_round(x)
    assert r[0] == 1.0
    assert r[1] == 1.0            # exactly half-way between 1.0 and 1.0078125 -> even (1.0)
    assert r[2] == 1.0078125 and r[3] == -1.0078125
    assert r[4] == 3.140625


def test_fp8_e4m3_rounding_model():
    """OCP e4m3 (bias 7, 3 mantissa bits, max 448, subnormal step 2^-9): known values, idempotence, monotonicity, clamp."""
    r = O.fp8_e4m3_round
    known = {0.3: 0.3125, 1.0: 1.0, 17.0: 16.0, 19.0: 20.0, 18.0: 18.0, 447.0: 448.0, 1e6: 448.0, -1e6: -448.0, 2.0 ** -10: 0.0,
             3 * 2.0 ** -10: 2.0 ** -8, 2.0 ** -9: 2.0 ** -9, 0.0146: 7 * 2.0 ** -9, -0.07: -0.0703125, 464.0: 448.0}
    for x, want in known.items():
        assert float(r(np.array([x]))[0]) == want, (x, float(r(np.array([x]))[0]), want)
    rng = np.random.default_rng(0)
    x = np.sort(rng.normal(0, 1, 5000) * np.exp2(rng.integers(-12, 9, 5000)))
    q = r(x)
    assert np.array_equal(r(q), q)                                  # idempotent
    assert np.all(np.diff(q) >= 0)                                  # monotone
    rel = np.abs(q - x)[np.abs(x) > 2.0 ** -6] / np.abs(x)[np.abs(x) > 2.0 ** -6]
    assert rel[np.abs(x[np.abs(x) > 2.0 ** -6]) < 448].max() <= 2.0 ** -4 + 1e-12   # half an ulp of a 3-bit mantissa
    assert len(np.unique(np.abs(q))) <= 127                          # 126 positive finite codes + zero


def test_discriminator_precision_modes_reduce_to_fp32_on_representable_operands():
    """With weights / embeddings that are exactly representable after the static scales, only the ACTIVATION and gradient
    classes are quantised: the first layer's output is identical in all modes; the modes differ afterwards."""
    rng = np.random.default_rng(5)
    I, hs = 60, (8, 12, 20, 16)
    D = O.init_discriminator(I, *hs, seed=2)
    for k in ("emb", "w1", "w2", "w3"):
        D[k] = (np.round(np.asarray(D[k], np.float64) * 256 / 16) * 16 / 256).astype(np.float32)    # multiples of 2^-4: exact in bf16 and e4m3 x 2^8
    pop, nic = rng.integers(0, I, 30), rng.integers(0, I, 30)
    masks = [np.ones((30, w)) for w in hs[1:]]
    T = {m: O.d_tower(D, pop, nic, masks, 1.0, dq=m) for m in (None, "bf16", "fp8")}
    assert np.array_equal(T[None]["tA"], T["bf16"]["tA"]) and np.array_equal(T[None]["tA"], T["fp8"]["tA"])
    assert not np.array_equal(T[None]["tC"], T["fp8"]["tC"])
    assert np.abs(T["bf16"]["y"] - T[None]["y"]).max() < 5e-3 and np.abs(T["fp8"]["y"] - T[None]["y"]).max() < 5e-2
    g = {m: O.d_tower_backward(D, T[m], masks, 1.0, T[m]["y"], dq=m) for m in (None, "bf16", "fp8")}
    for k in O.D_KEYS:
        den = np.abs(g[None][k]).max()
        assert np.abs(g["bf16"][k] - g[None][k]).max() < 2e-2 * den and np.abs(g["fp8"][k] - g[None][k]).max() < 0.3 * den, k
