"""CPU ORACLE (test infrastructure only) for the Long-Tail-GAN adversarial training path.

This file is a numpy restatement of the reference's algorithm.  It is NOT part of the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
path (long-tail-gan_amd/) never imports anything under oracle/ and fails loudly without its HIP
extension.

Parity status
-------------
* Index path / sampler / metrics restatements are PINNED against the reference's own modules
  (data_processing.py, sample.py, eval_functions.py imported read-only in the build container;
  fixtures + generating script under tests/golden/).
* Model math (MultiVAE / discriminator / losses / Adam) follows the cited reference lines but the
  arithmetic lives in TensorFlow 1.x (unpinned version, not vendored, not installable here):
  "parity unpinned" for TF library behaviour (matmul accumulation order, tf.nn.dropout mask
  convention, AdamOptimizer's shared beta-power accumulators).  The restatement is cross-checked
  against torch-CPU autograd in tests/test_oracle_math.py.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# counter-based RNG shared bit-for-bit with the HIP kernels (csrc/ltg_rng.h)
# ----------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_SC = np.uint64(0xD6E8FEB86659FD93)

# stream ids (must match csrc/ltg_rng.h)
STREAM_VAE_DROPOUT = 1
STREAM_VAE_EPS = 2
STREAM_D_DROP_A = 3   # branch layer popular  (h1)
STREAM_D_DROP_B = 4   # branch layer niche    (h2)
STREAM_D_DROP_C = 5   # fc layer              (h3)
STREAM_GUMBEL = 6
STREAM_POP_PICK = 7


def rng_u64(seed, stream, step, idx):
    """splitmix64-style hash of (seed, stream, step, idx) -> uint64.  idx may be an array."""
    with np.errstate(over="ignore"):
        idx = np.asarray(idx, dtype=np.uint64)
        z = (np.uint64(seed) ^ (np.uint64(stream) * _SC)) + np.uint64(step) * _C2
        z = z + (idx + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def rng_uniform(seed, stream, step, idx):
    """uniform in [0,1) with 24 bits, exactly representable in fp32."""
    z = rng_u64(seed, stream, step, idx)
    return ((z >> np.uint64(40)).astype(np.float64)) * (1.0 / 16777216.0)


def rng_normal(seed, stream, step, idx):
    """Box-Muller from two 24-bit uniforms (u1 in (0,1])."""
    z = rng_u64(seed, stream, step, idx)
    u1 = (((z >> np.uint64(40)).astype(np.float64)) + 1.0) * (1.0 / 16777216.0)
    u2 = (((z >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)) * (1.0 / 16777216.0)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, same as v_cvt_pk_bf16_f32 for finite values)
# ----------------------------------------------------------------------------------------------
def bf16_round(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def _q(x, quant):
    return bf16_round(x).astype(x.dtype) if quant else x


def fp8_e4m3_round(x):
    """Round to the nearest OCP e4m3 value (3 mantissa bits, normals 2^-6 .. 448, subnormal step 2^-9), ties to even;
    the hardware path clamps to +-448 (the largest finite e4m3) first (ltg_gemm.h: ltg_f2fp8), so does this."""
    x = np.clip(np.asarray(x, dtype=np.float64), -448.0, 448.0)
    _, e = np.frexp(x)                                   # |x| = m * 2^e, m in [0.5, 1)
    step = np.exp2(np.maximum(e - 4, -9).astype(np.float64))
    return np.rint(x / step) * step


# static scale exponents of the fp8 discriminator mode (csrc/ltg_kernels.hip: FP8_S_*)
FP8_S = {"emb": 8, "w": 8, "act": 6, "g3": 8, "g1": 7}


def _dq(x, mode, cls):
    """operand quantisation of one discriminator GEMM: mode None (fp32) | "bf16" | "fp8" (static scale of class cls)."""
    if mode is None:
        return x
    if mode == "bf16":
        return bf16_round(np.asarray(x, np.float32)).astype(x.dtype)
    sc = float(1 << FP8_S[cls])
    return (fp8_e4m3_round(np.asarray(x, np.float64) * sc) / sc).astype(x.dtype)


# ----------------------------------------------------------------------------------------------
# MultiVAE generator  (Codes/Base_Recommender/MultiVAE.py:95-230, Codes/generator.py:4-22)
# ----------------------------------------------------------------------------------------------
H_ENC = 600
Z_DIM = 200


def xavier_uniform(rng, fan_in, fan_out, shape=None):
    """tf.contrib.layers.xavier_initializer (uniform): U(+-sqrt(6/(fan_in+fan_out))).
    MultiVAE.py:199-202.  (TF's RNG stream itself is not reproducible here: parity unpinned.)"""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape or (fan_in, fan_out)).astype(np.float32)


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: re-draw values beyond 2 sigma. MultiVAE.py:204-207, discriminator.py:14."""
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_generator(n_items, seed=98765):
    """Parameter set in the reference's order: [W_q0, W_q1, W_p0, W_p1, b_q0, b_q1, b_p0, b_p1]
    (MultiVAE.py:129-141) with TF shapes ([in, out]).  p_dims=[200,600,I] (generator.py:13)."""
    rng = np.random.default_rng(seed)
    I = n_items
    return {
        "Wq0": xavier_uniform(rng, I, H_ENC),
        "Wq1": xavier_uniform(rng, H_ENC, 2 * Z_DIM),
        "Wp0": xavier_uniform(rng, Z_DIM, H_ENC),
        "Wp1": xavier_uniform(rng, H_ENC, I),
        "bq0": truncated_normal(rng, (H_ENC,), 1e-3),
        "bq1": truncated_normal(rng, (2 * Z_DIM,), 1e-3),
        "bp0": truncated_normal(rng, (H_ENC,), 1e-3),
        "bp1": truncated_normal(rng, (I,), 1e-3),
    }


G_KEYS = ["Wq0", "Wq1", "Wp0", "Wp1", "bq0", "bq1", "bp0", "bp1"]


def vae_forward(P, X, drop_mask, keep, eps, is_training, anneal, dtype=np.float64, quant=False):
    """Forward pass of MultiVAE.build_graph (MultiVAE.py:104-186).

    X          [B,I] dense multi-hot (float)
    drop_mask  [B,I] {0,1}: tf.nn.dropout's keep indicator (floor(keep+u)); MultiVAE.py:149
    eps        [B,200]: tf.random_normal of MultiVAE.py:178
    quant      True: round the decoder-GEMM operands (h2, W_p1) to bf16 like the HIP bf16 path.
    Returns a dict of every intermediate.
    """
    f = lambda a: np.asarray(a, dtype=dtype)
    X = f(X)
    B = X.shape[0]
    nrm = np.sqrt(np.maximum((X * X).sum(1, keepdims=True), 1e-12))       # l2_normalize :148
    h = X / nrm
    h = h / dtype(keep) * f(drop_mask)                                    # dropout :149
    a1 = h @ f(P["Wq0"]) + f(P["bq0"])                                    # :152
    h1 = np.tanh(a1)                                                      # :155
    a2 = h1 @ f(P["Wq1"]) + f(P["bq1"])
    mu, logvar = a2[:, :Z_DIM], a2[:, Z_DIM:]                             # :157-158
    std = np.exp(0.5 * logvar)                                            # :160
    KL_rows = (0.5 * (-logvar + np.exp(logvar) + mu ** 2 - 1)).sum(1)     # :161-162
    KL = KL_rows.mean()
    z = mu + dtype(is_training) * f(eps) * std                            # :180-181
    h2 = np.tanh(z @ f(P["Wp0"]) + f(P["bp0"]))                           # :168-172
    logits = _q(h2, quant) @ _q(f(P["Wp1"]), quant) + f(P["bp1"])         # :169
    mx = logits.max(1, keepdims=True)
    lse = mx + np.log(np.exp(logits - mx).sum(1, keepdims=True))
    logsm = logits - lse                                                  # :108
    neg_ll_rows = -(logsm * X).sum(1)
    neg_ll = neg_ll_rows.mean()                                           # :110-112
    neg_elbo = neg_ll + dtype(anneal) * KL                                # :119 (lam=0)
    probs = np.exp(logsm)                                                 # :143 softmax
    return dict(h=h, h1=h1, mu=mu, logvar=logvar, std=std, KL=KL, KL_rows=KL_rows, z=z, h2=h2,
                logits=logits, lse=lse[:, 0], probs=probs, neg_ll=neg_ll, neg_ll_rows=neg_ll_rows,
                neg_elbo=neg_elbo, nrm=nrm[:, 0])


# ----------------------------------------------------------------------------------------------
# discriminator (Codes/discriminator.py:3-58)
# ----------------------------------------------------------------------------------------------
D_KEYS = ["w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4"]


def init_discriminator(feature_len, h0, h1, h2, h3, seed=0):
    rng = np.random.default_rng(seed)
    return {
        "emb": truncated_normal(rng, (feature_len, h0), 0.1),   # discriminator.py:14 (frozen, Q4)
        "w1": truncated_normal(rng, (h0, h1), 0.1), "b1": np.zeros(h1, np.float32),   # :23-24
        "w2": truncated_normal(rng, (h0, h2), 0.1), "b2": np.zeros(h2, np.float32),   # :28-29
        "w3": truncated_normal(rng, (h1 + h2, h3), 0.1), "b3": np.zeros(h3, np.float32),  # :36-37
        "w4": truncated_normal(rng, (h3, 1), 0.1), "b4": np.zeros(1, np.float32),     # :40-41
    }


def d_tower(D, pop_ids, niche_ids, masks, keep, dtype=np.float64, dq=None):
    """t(a,b) of discriminator.py:16-19,25,30,33,44-45 (same code serves :51-55).
    masks = (mA [K,h1], mB [K,h2], mC [K,h3]) keep indicators.  dq: operand precision of the three GEMMs of the
    ltg_config.d_precision modes (None = fp32 like the reference, "bf16", "fp8"); the output layer is always fp32."""
    f = lambda a: np.asarray(a, dtype=dtype)
    mA, mB, mC = (f(m) for m in masks)
    ea = f(D["emb"])[pop_ids]
    eb = f(D["emb"])[niche_ids]
    tA = np.tanh(_dq(ea, dq, "emb") @ _dq(f(D["w1"]), dq, "w") + f(D["b1"]))
    tB = np.tanh(_dq(eb, dq, "emb") @ _dq(f(D["w2"]), dq, "w") + f(D["b2"]))
    aA = tA / dtype(keep) * mA
    aB = tB / dtype(keep) * mB
    hin = np.concatenate([aA, aB], 1)
    if dq is not None:
        hin = hin.astype(np.float32).astype(dtype)      # activations live in fp32 buffers between the kernels
    tC = np.tanh(_dq(hin, dq, "act") @ _dq(f(D["w3"]), dq, "w") + f(D["b3"]))
    aC = tC / dtype(keep) * mC
    s = aC @ f(D["w4"]) + f(D["b4"])
    y = 1.0 / (1.0 + np.exp(-s))
    return dict(ea=ea, eb=eb, tA=tA, tB=tB, hin=hin, tC=tC, aC=aC, s=s[:, 0], y=y[:, 0])


def d_tower_backward(D, T, masks, keep, ds, dtype=np.float64, dq=None, dact16=True):
    """gradients of sum(ds * s) wrt the 8 trainable tensors (emb is frozen: discriminator.py:47).  dq as in d_tower:
    the bias gradients of the quantised GEMMs are column sums of the QUANTISED operand (ones-augmented row).
    dact16 (fp8 mode only): round d a / d pre of the branch layers to bf16 as the device stores it (dA1T_16).  False = the same
    pipeline WITHOUT that storage rounding: the independent variant tests pin the approximation's cost against
    (tests/test_gpu_parity.py::test_d_step_precision_modes)."""
    f = lambda a: np.asarray(a, dtype=dtype)
    mA, mB, mC = (f(m) for m in masks)
    ds = f(ds)[:, None]
    g = {}
    g["w4"] = T["aC"].T @ ds
    g["b4"] = ds.sum(0)
    daC = ds @ f(D["w4"]).T
    dpC = daC * mC / dtype(keep) * (1 - T["tC"] ** 2)
    if dq is not None:
        dpC = dpC.astype(np.float32).astype(dtype)
    qC = _dq(dpC, dq, "g3")
    g["w3"] = _dq(T["hin"], dq, "act").T @ qC
    g["b3"] = qC.sum(0)
    dhin = qC @ _dq(f(D["w3"]), dq, "w").T
    h1 = mA.shape[1]
    fA, fB = mA / dtype(keep) * (1 - T["tA"] ** 2), mB / dtype(keep) * (1 - T["tB"] ** 2)
    if dq == "fp8" and dact16:     # the fp8 mode keeps d a / d pre of the branch layers as bf16 (csrc/ltg_fp8bwd.h: dA1T_16), not an fp32 copy of a
        fA, fB = bf16_round(fA.astype(np.float32)).astype(dtype), bf16_round(fB.astype(np.float32)).astype(dtype)
    dpA = dhin[:, :h1] * fA
    dpB = dhin[:, h1:] * fB
    if dq is not None:
        dpA, dpB = dpA.astype(np.float32).astype(dtype), dpB.astype(np.float32).astype(dtype)
    qA, qB = _dq(dpA, dq, "g1"), _dq(dpB, dq, "g1")
    g["w1"] = _dq(T["ea"], dq, "emb").T @ qA
    g["b1"] = qA.sum(0)
    g["w2"] = _dq(T["eb"], dq, "emb").T @ qB
    g["b2"] = qB.sum(0)
    return g


def d_loss_and_grads(D, real, fake, keep, dtype=np.float64):
    """d_loss = -sum log y_data - sum log(1-y_generated)  (train.py:142).
    real = (pop_n, niche, masks3), fake = (pop_g, gen, masks3)."""
    Tr = d_tower(D, real[0], real[1], real[2], keep, dtype)
    Tf = d_tower(D, fake[0], fake[1], fake[2], keep, dtype)
    loss = -np.log(Tr["y"]).sum() - np.log(1.0 - Tf["y"]).sum()
    gr = d_tower_backward(D, Tr, real[2], keep, -(1.0 - Tr["y"]), dtype)
    gf = d_tower_backward(D, Tf, fake[2], keep, Tf["y"], dtype)
    g = {k: gr[k] + gf[k] for k in gr}
    return loss, g, Tr, Tf


# ----------------------------------------------------------------------------------------------
# generator loss + closed-form gradients (train.py:145-157)
# ----------------------------------------------------------------------------------------------
def g_loss_and_grads(P, X, drop_mask, keep, eps, anneal, lam, S_rows, S_cols, cnt, sum_y,
                     is_training=1.0, dtype=np.float64, quant=False):
    """g_loss = neg_ELBO - (lam/cnt) * (sum_S p) * (sum_j y_j)   (train.py:155; Q2 broadcast).

    S_rows/S_cols: positions of the non-zero mask entries ('generated_tags', train.py:132,145).
    sum_y: sum of y_generated over the fake pairs (discriminator forward, dropout on).
    Returns (losses dict, grads dict in G_KEYS, forward dict).
    """
    f = lambda a: np.asarray(a, dtype=dtype)
    F = vae_forward(P, X, drop_mask, keep, eps, is_training, anneal, dtype, quant)
    X = f(X)
    B = X.shape[0]
    p = F["probs"]
    S_rows = np.asarray(S_rows, dtype=np.int64)
    S_cols = np.asarray(S_cols, dtype=np.int64)
    sum_p = p[S_rows, S_cols].sum() if len(S_rows) else dtype(0)
    c = dtype(lam) / dtype(cnt) * dtype(sum_y)
    gan_loss = -c * sum_p                                                 # train.py:157
    g_loss = F["neg_elbo"] + gan_loss                                     # train.py:155
    # d/dlogits
    n_b = X.sum(1, keepdims=True)
    Pb = np.zeros((B, 1), dtype)
    np.add.at(Pb[:, 0], S_rows, p[S_rows, S_cols])
    ind = np.zeros_like(p)
    ind[S_rows, S_cols] = 1.0
    dlog = (p * n_b - X) / B - c * p * (ind - Pb)
    dlq = _q(dlog, quant)
    h2q = _q(F["h2"], quant)
    g = {}
    g["Wp1"] = h2q.T @ dlq
    g["bp1"] = dlq.sum(0)   # the HIP path takes it from the ones-augmented column of the same (bf16) GEMM
    dh2 = dlq @ _q(f(P["Wp1"]), quant).T
    da2 = dh2 * (1 - F["h2"] ** 2)
    g["Wp0"] = F["z"].T @ da2
    g["bp0"] = da2.sum(0)
    dz = da2 @ f(P["Wp0"]).T
    dmu = dz + dtype(anneal) * F["mu"] / B
    dlv = dz * dtype(is_training) * f(eps) * F["std"] * 0.5 + dtype(anneal) * 0.5 * (np.exp(F["logvar"]) - 1) / B
    dmlv = np.concatenate([dmu, dlv], 1)
    g["Wq1"] = F["h1"].T @ dmlv
    g["bq1"] = dmlv.sum(0)
    dh1 = dmlv @ f(P["Wq1"]).T
    da1 = dh1 * (1 - F["h1"] ** 2)
    g["Wq0"] = F["h"].T @ da1
    g["bq0"] = da1.sum(0)
    losses = dict(g_loss=g_loss, vae_loss=F["neg_elbo"], gan_loss=gan_loss, sum_p=sum_p, c=c)
    F.update(dlog=dlog, dh2=dh2, da2=da2, dz=dz, dmlv=dmlv, dh1=dh1, da1=da1)
    return losses, g, F


# ----------------------------------------------------------------------------------------------
# TF1 AdamOptimizer with ONE shared step counter (train.py:160-164; Q5)
# ----------------------------------------------------------------------------------------------
class SharedAdam:
    """tf.train.AdamOptimizer(lr): m,v per variable; beta1_power/beta2_power per optimizer, so the
    bias-correction step t advances on every D step and every G step (Q5).
    theta -= lr_t * m / (sqrt(v) + eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t)."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}
        self.dtype = dtype

    def lr_t(self, t=None):
        t = self.t if t is None else t
        return self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def apply(self, params, grads, keys):
        self.t += 1
        lr_t = self.dtype(self.lr_t())
        dt = self.dtype
        for k in keys:
            gk = np.asarray(grads[k], dtype=dt).reshape(params[k].shape)
            if k not in self.m:
                self.m[k] = np.zeros(params[k].shape, dt)
                self.v[k] = np.zeros(params[k].shape, dt)
            self.m[k] = dt(self.b1) * self.m[k] + dt(1 - self.b1) * gk
            self.v[k] = dt(self.b2) * self.v[k] + dt(1 - self.b2) * gk * gk
            params[k] = (np.asarray(params[k], dt) - lr_t * self.m[k] / (np.sqrt(self.v[k]) + dt(self.eps))).astype(params[k].dtype)


def anneal_value(update_count, total_anneal_steps=20000, anneal_cap=0.2):
    """train.py:319-322 (update_count is incremented AFTER this is evaluated, :324)."""
    if total_anneal_steps > 0:
        return min(anneal_cap, 1.0 * update_count / total_anneal_steps)
    return anneal_cap


# ----------------------------------------------------------------------------------------------
# sampler (Codes/sample.py:40-67) and fake-pair construction (Codes/train.py:212-251)
# ----------------------------------------------------------------------------------------------
def legacy_choice_no_replace(u_stream, p, size):
    """numpy RandomState.choice(replace=False, p=p) restated with injected uniforms
    (numpy/random/mtrand.pyx 'choice', legacy algorithm).  u_stream: callable n -> n uniforms.
    Used to pin the restatement of sample.py:54 against np.random under a fixed seed."""
    p = np.array(p, dtype=np.float64)
    pop = p.shape[0]
    if np.count_nonzero(p > 0) < size:
        raise ValueError("Fewer non-zero entries in p than size")
    n_uniq = 0
    found = np.zeros(size, dtype=np.int64)
    while n_uniq < size:
        x = u_stream(size - n_uniq)
        if n_uniq > 0:
            p[found[0:n_uniq]] = 0
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        new = cdf.searchsorted(x, side="right")
        _, unique_indices = np.unique(new, return_index=True)
        unique_indices.sort()
        new = new.take(unique_indices)
        found[n_uniq:n_uniq + new.size] = new
        n_uniq += new.size
    return found


def gumbel_keys(p_cand, u, dtype=np.float64):
    """Gumbel-top-k key: log p - log(-log u).  Successive sampling without replacement
    (what sample.py:54 draws) == taking the k largest keys (Plackett-Luce)."""
    p = np.asarray(p_cand, dtype=dtype)
    u = np.asarray(u, dtype=dtype)
    with np.errstate(divide="ignore"):
        key = np.log(p) - np.log(-np.log(np.maximum(u, 2.0 ** -25)))
    return np.where(p > 0, key, -np.inf)


def sample_user(cand, p_cand, k, u, dtype=np.float64):
    """sample_from_generator_new (sample.py:40-67) with Gumbel noise u[len(cand)] injected.
    K_eff = min(k, nnz(p)) (Q10: the exception-driven decrement); all-zero p -> empty sample
    (documented divergence: the reference crashes).  Returns ascending ids (train.py:230 sorts)."""
    cand = np.asarray(cand)
    p = np.asarray(p_cand, dtype=dtype)
    k_eff = min(int(k), int(np.count_nonzero(p > 0)))
    if k_eff <= 0:
        return np.zeros(0, dtype=np.int64)
    key = gumbel_keys(p, u, dtype)
    # rank = number of strictly larger keys (+ earlier index among equals) -- same rule as the kernel
    order = np.lexsort((np.arange(len(cand)), -key))
    sel = np.sort(order[:k_eff])
    return cand[sel].astype(np.int64)


def build_fake_pairs(gen_ids, pop_list, u_pick, valid_item):
    """train.py:232-248 for one user: each sampled niche item is paired with a uniformly random
    popular item of the user (np.random.choice(range(n)), :236); pairs touching an item outside
    ITEM_FEATURE_DICT are dropped (:240-243, Q9).  u_pick[j] in [0,1) -> index floor(u*n).
    Returns (x_generated, x_popular_g, kept_flags)."""
    n = len(pop_list)
    xg, xp, kept = [], [], []
    for j, gid in enumerate(gen_ids):
        pidx = min(int(u_pick[j] * n), n - 1)
        pop = pop_list[pidx]
        ok = bool(valid_item[gid]) and bool(valid_item[pop])
        kept.append(ok)
        if ok:
            xg.append(int(gid))
            xp.append(int(pop))
    return xg, xp, kept


# ----------------------------------------------------------------------------------------------
# metrics (Codes/eval_functions.py:11-62; masking train.py:341)
# ----------------------------------------------------------------------------------------------
def ndcg_binary_at_k(pred, heldout_dense, k=100):
    """NDCG_binary_at_k_batch restated without bottleneck (eval_functions.py:11-38).
    Ties are broken by lower item index first (the kernel's rule); the reference's tie order is
    whatever bn.argpartition yields (only matters for exactly equal scores).
    Returns the list over users with IDCG != 0, like the reference."""
    pred = np.asarray(pred)
    n_users, n_items = pred.shape
    kk = min(k, n_items)
    idx = np.lexsort((np.broadcast_to(np.arange(n_items), pred.shape), -pred), axis=1)[:, :kk]
    tp = 1.0 / np.log2(np.arange(2, kk + 2))
    held = np.asarray(heldout_dense) > 0
    DCG = (held[np.arange(n_users)[:, None], idx] * tp).sum(1)
    nnz = held.sum(1)
    IDCG = np.array([tp[:min(n, kk)].sum() for n in nnz])
    return [DCG[i] / IDCG[i] for i in range(n_users) if IDCG[i] != 0]


def recall_at_k(pred, heldout_dense, k):
    """Recall_at_k_batch (eval_functions.py:40-62)."""
    pred = np.asarray(pred)
    n_users, n_items = pred.shape
    idx = np.lexsort((np.broadcast_to(np.arange(n_items), pred.shape), -pred), axis=1)[:, :k]
    held = np.asarray(heldout_dense) > 0
    hit = held[np.arange(n_users)[:, None], idx].sum(1).astype(np.float32)
    denom = np.minimum(k, held.sum(1))
    return [hit[i] / denom[i] for i in range(n_users) if denom[i] != 0]
