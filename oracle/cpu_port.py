"""CPU BASELINE PORT (test/bench infrastructure only) -- a faithful torch-CPU fp32 restatement of the
reference's training loop with the reference's own structure: dense [B,I] float32 feeds, dense
mask feed, op-by-op graph (no fusion), autograd for the gradients, one shared Adam, and the
per-user Python sampling loop of train.py:212-251 calling a sample.py:40-67 restatement that uses
np.random.choice exactly like the reference.  Used ONLY by bench.py's `cpu_baseline` leg (kind
"port": TensorFlow 1.x cannot be installed, so the TF-CPU reference itself cannot be timed).

Never imported by the product path.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def sample_from_generator_new(elements, probabilities_li, to_sample, num_elements):
    """sample.py:40-67 (same control flow, np.random global stream)."""
    sampled_bin = np.zeros([num_elements], dtype=float)
    p = np.asarray(probabilities_li)
    if p.sum() != 0.0:
        p = p / (1.0 * p.sum())
    else:
        return sampled_bin, np.zeros(0, dtype=np.int64)   # the reference crashes here (Q10)
    sampled = np.zeros(0, dtype=np.int64)
    while True:
        try:
            sampled = np.random.choice(elements, to_sample, p=p, replace=False)
            break
        except Exception:
            to_sample -= 1
            if to_sample == 0:
                break
    sampled_bin[sampled] = 1
    return sampled_bin, np.asarray(sampled)


class TfAdam:
    """tf.train.AdamOptimizer semantics incl. the shared beta-power step (Q5)."""

    def __init__(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.state = {}

    @torch.no_grad()
    def step(self, params):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p in params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), (torch.zeros_like(p), torch.zeros_like(p)))
            m, v = st
            m.mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)
            p.grad = None


class CpuPort:
    def __init__(self, idx, h=(100, 150, 250, 300), lr=1e-4, batch_size=100, seed=0, threads=None, init=None):
        """init: optional (generator arrays in TF shapes [W_q0 [I,600], W_q1, W_p0, W_p1 [600,I], b_q0, b_q1, b_p0, b_p1],
        emb [I,h0], discriminator arrays [w1, b1, w2, b2, w3, b3, w4 [h3,1], b4]) -- so that a device run can start from the
        very same variables (tests/golden/make_ndcg_gate.py)."""
        if threads:
            torch.set_num_threads(threads)
        self.idx, self.BS = idx, batch_size
        I = idx.n_items
        g = torch.Generator().manual_seed(seed)
        if init is not None:
            ga, emb, da = init
            f = lambda a: torch.tensor(np.asarray(a, dtype=np.float32)).requires_grad_()
            self.g_params = [f(a) for a in ga]
            self.emb = torch.tensor(np.asarray(emb, dtype=np.float32))
            self.d_params = [f(a) for a in da]
            self.opt = TfAdam(lr)
            self.update_count = 0.0
            self.valid = set(np.nonzero(idx.valid_item)[0].tolist())
            return
        xav = lambda a, b: ((torch.rand(a, b, generator=g) * 2 - 1) * np.sqrt(6.0 / (a + b))).requires_grad_()
        tn = lambda *s, std: (torch.randn(*s, generator=g).clamp_(-2, 2) * std).requires_grad_()
        self.g_params = [xav(I, 600), xav(600, 400), xav(200, 600), xav(600, I), tn(600, std=1e-3), tn(400, std=1e-3),
                         tn(600, std=1e-3), tn(I, std=1e-3)]
        h0, h1, h2, h3 = h
        self.emb = torch.randn(I, h0, generator=g).clamp_(-2, 2) * 0.1
        self.d_params = [tn(h0, h1, std=0.1), torch.zeros(h1, requires_grad=True), tn(h0, h2, std=0.1),
                         torch.zeros(h2, requires_grad=True), tn(h1 + h2, h3, std=0.1), torch.zeros(h3, requires_grad=True),
                         tn(h3, 1, std=0.1), torch.zeros(1, requires_grad=True)]
        self.opt = TfAdam(lr)
        self.update_count = 0.0
        self.valid = set(np.nonzero(idx.valid_item)[0].tolist())

    # --- graph pieces (MultiVAE.py:145-186, discriminator.py:16-55)
    def vae(self, X, keep, is_training, anneal):
        Wq0, Wq1, Wp0, Wp1, bq0, bq1, bp0, bp1 = self.g_params
        h = torch.nn.functional.normalize(X, dim=1, eps=1e-12)
        h = torch.nn.functional.dropout(h, 1 - keep, training=True)
        h1 = torch.tanh(h @ Wq0 + bq0)
        a2 = h1 @ Wq1 + bq1
        mu, lv = a2[:, :200], a2[:, 200:]
        std = torch.exp(0.5 * lv)
        KL = (0.5 * (-lv + torch.exp(lv) + mu ** 2 - 1)).sum(1).mean()
        z = mu + is_training * torch.randn_like(std) * std
        logits = torch.tanh(z @ Wp0 + bp0) @ Wp1 + bp1
        neg_ll = -(torch.log_softmax(logits, 1) * X).sum(-1).mean()
        return torch.softmax(logits, 1), neg_ll + anneal * KL

    def tower(self, pop, nic, keep):
        w1, b1, w2, b2, w3, b3, w4, b4 = self.d_params
        dr = lambda t: torch.nn.functional.dropout(t, 1 - keep, training=True)
        a = dr(torch.tanh(self.emb[pop] @ w1 + b1))
        b = dr(torch.tanh(self.emb[nic] @ w2 + b2))
        c = dr(torch.tanh(torch.cat([a, b], 1) @ w3 + b3))
        return torch.sigmoid(c @ w4 + b4)

    @torch.no_grad()
    def validate(self, tr_csr, te_csr, chunk=2000):
        """train.py:333-348: forward with dropout ON (keep_prob_ph is never fed, Q3), fold-in items -> -inf, NDCG@100 /
        Recall@20 / Recall@50 (eval_functions.py:11-62 restated in oracle/ltg_oracle.py)."""
        from . import ltg_oracle as O
        nd, r20, r50 = [], [], []
        for lo in range(0, tr_csr.shape[0], chunk):
            Xn = np.asarray(tr_csr[lo:lo + chunk].toarray(), dtype=np.float32)
            probs, _ = self.vae(torch.from_numpy(Xn), 0.75, 0.0, 1.0)
            pred = probs.numpy().astype(np.float64)
            pred[Xn > 0] = -np.inf
            held = np.asarray(te_csr[lo:lo + chunk].toarray())
            nd.append(O.ndcg_binary_at_k(pred, held, 100))
            r20.append(O.recall_at_k(pred, held, 20))
            r50.append(O.recall_at_k(pred, held, 50))
        return float(np.mean(np.concatenate(nd))), float(np.mean(np.concatenate(r20))), float(np.mean(np.concatenate(r50)))

    # --- one "mini epoch" over batches [b0, b1): phases C, D x S, G x S (train.py:192-329)
    def run(self, b0, b1, S):
        idx, BS, I = self.idx, self.BS, self.idx.n_items
        tr = idx.train
        cache = []
        t0 = time.perf_counter()
        for b in range(b0, b1):
            lo, hi = b * BS, min(idx.N, (b + 1) * BS)
            X = torch.from_numpy(tr[lo:hi].toarray().astype("float32"))
            with torch.no_grad():
                probs, _ = self.vae(X, 0.75, 0.0, 1.0)
            probs = probs.numpy()
            xn, xpn, xg, xpg, tags, cnt = [], [], [], [], [], 0
            for ii, u in enumerate(range(lo, hi)):
                if not idx.user_ok[u]:
                    tags.append([0] * I)
                    continue
                pops = idx.pop_idx[idx.pop_ptr[u]:idx.pop_ptr[u + 1]]
                cand = idx.cand_idx[idx.cand_ptr[u]:idx.cand_ptr[u + 1]]
                xn += idx.real_nic[idx.real_ptr[u]:idx.real_ptr[u + 1]].tolist()
                xpn += idx.real_pop[idx.real_ptr[u]:idx.real_ptr[u + 1]].tolist()
                binv, ids = sample_from_generator_new(cand, probs[ii, cand], int(idx.n_sample[u]), I)
                ids = np.sort(ids)
                for gid in ids:
                    pid = pops[np.random.choice(range(len(pops)))]
                    if gid not in self.valid or pid not in self.valid:
                        binv[gid] = 0
                        continue
                    xg.append(int(gid))
                    xpg.append(int(pid))
                    cnt += 1
                tags.append(binv)
            if not xg:
                continue
            cache.append((X, torch.from_numpy(np.asarray(tags, dtype=np.float32)), torch.tensor(xg), torch.tensor(xpg),
                          torch.tensor(xn), torch.tensor(xpn), float(cnt)))
        t1 = time.perf_counter()
        order = np.arange(len(cache))
        np.random.shuffle(order)
        for _ in range(S):
            for k in order:
                X, M, xg, xpg, xn, xpn, cnt = cache[k]
                y_d = self.tower(xpn, xn, 0.7)
                y_g = self.tower(xpg, xg, 0.7)
                d_loss = -torch.log(y_d).sum() - torch.log(1 - y_g).sum()
                d_loss.backward()
                self.opt.step(self.d_params)
        t2 = time.perf_counter()
        for _ in range(S):
            for k in order:
                X, M, xg, xpg, xn, xpn, cnt = cache[k]
                anneal = min(0.2, self.update_count / 20000.0)
                self.update_count += 1
                probs, vae_loss = self.vae(X, 0.75, 1.0, anneal)
                with torch.no_grad():
                    y_g = self.tower(xpg, xg, 0.7)
                s = (probs * M).reshape(-1)
                nz = s[s != 0]
                g_loss = vae_loss - (1.0 / cnt) * (nz * y_g).sum()      # [K]*[K,1] broadcast (Q2)
                g_loss.backward()
                for p in self.d_params:
                    p.grad = None
                self.opt.step(self.g_params)
        t3 = time.perf_counter()
        users = min(idx.N, b1 * BS) - b0 * BS
        return dict(users=users, t_create=t1 - t0, t_d=t2 - t1, t_g=t3 - t2, t_total=t3 - t0)


def time_cpu_baseline(idx, budget_s=20.0, S=10, batch_size=100):
    """Runs mini-epochs over a bounded number of batches (~budget_s of CPU work).  The thread count is the
    best of a short calibration (torch's default of one thread per core is far from optimal for these small
    GEMMs on a many-core host); `cores` reports the threads actually used."""
    import os
    ncpu = os.cpu_count() or 1
    port = CpuPort(idx, batch_size=batch_size)
    nb_total = (idx.N + batch_size - 1) // batch_size
    port.run(0, 1, 1)                       # warm-up
    best_t, best_dt = None, None
    for t in sorted({min(ncpu, x) for x in (8, 16, 32, 64)}):
        torch.set_num_threads(t)
        r = port.run(0, 1, 2)
        if best_dt is None or r["t_total"] < best_dt:
            best_t, best_dt = t, r["t_total"]
    torch.set_num_threads(best_t)
    per_batch = max(best_dt * S / 2.0, 1e-3)
    nb = int(max(1, min(nb_total - 1, budget_s / per_batch)))
    r = port.run(1, 1 + nb, S)
    return dict(value=r["users"] / r["t_total"], unit="users/s", cores=best_t, kind="port",
                sample="%d of %d batches (%d users) through C + %dxD + %dxG on torch-CPU fp32 (%d threads of %d cores), %.1f s" %
                       (nb, nb_total, r["users"], S, S, best_t, ncpu, r["t_total"]),
                phases_s={k: r[k] for k in ("t_create", "t_d", "t_g")})
