"""ORACLE training trajectory (test infrastructure only): the global-epoch loop of the reference
(Codes/train.py:180-356) in fp64 numpy, driven by the SAME counter RNG streams and the same call
sequence as ltgan.trainer.Trainer, so that a GPU run can be compared step for step
(tests/test_gpu_trajectory.py, fixture tests/golden/oracle_trajectory.npz).
"""
from __future__ import annotations

import numpy as np

from . import ltg_oracle as O


class OracleTrainer:
    def __init__(self, idx, P, D, hs, lr=1e-4, S=10, lam=1.0, batch_size=100, seed=98765, quant=True, shuffle_seed=0,
                 n_batches=None):
        self.idx, self.P, self.D, self.hs = idx, {k: np.asarray(v, np.float64) for k, v in P.items()}, \
            {k: np.asarray(v, np.float64) for k, v in D.items()}, hs
        self.S, self.lam, self.BS, self.seed, self.quant = S, lam, batch_size, seed, quant
        self.adam = O.SharedAdam(lr)
        self.update_count = 0.0
        self.rng_step = 0
        self.np_rng = np.random.RandomState(shuffle_seed)
        nb = (idx.N + batch_size - 1) // batch_size
        self.n_batches = nb if n_batches is None else min(nb, n_batches)
        self.I = idx.n_items
        self.fake = {}
        self.d_log, self.g_log = [], []      # every step of the LAST phase: d_loss / (g_loss, vae_loss, gan_loss, anneal) (train.py:300, :326)

    def _step(self):
        self.rng_step += 1
        return self.rng_step

    def _rows(self, b):
        return b * self.BS, min(self.idx.N, (b + 1) * self.BS)

    def _X(self, b):
        lo, hi = self._rows(b)
        return np.asarray(self.idx.train[lo:hi].toarray(), np.float64)

    def _mask(self, step, B):
        idx = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(self.I) + np.arange(self.I, dtype=np.uint64)[None, :])
        return (O.rng_uniform(self.seed, O.STREAM_VAE_DROPOUT, step, idx).astype(np.float32) < np.float32(0.75)).astype(np.float64)

    # ------------------------------------------------------------------ phase C
    def create_phase(self, fake_override=None):
        idx, I = self.idx, self.I
        self.fake = {}
        for b in range(self.n_batches):
            st = self._step()
            lo, hi = self._rows(b)
            B = hi - lo
            if fake_override is not None:
                self.fake[b] = fake_override[b]
                continue
            X = self._X(b)
            F = O.vae_forward(self.P, X, self._mask(st, B), 0.75, np.zeros((B, O.Z_DIM)), 0.0, 1.0, np.float64, self.quant)
            rows, gen, pop, pos = [], [], [], []
            s0 = int(idx.slot_ptr[lo])
            for r in range(B):
                u = lo + r
                ns = int(idx.n_sample[u])
                if ns == 0:
                    continue
                cand = idx.cand_idx[idx.cand_ptr[u]:idx.cand_ptr[u + 1]].astype(np.int64)
                p = F["probs"][r, cand]
                ug = O.rng_uniform(self.seed, O.STREAM_GUMBEL, st, np.uint64(r) * np.uint64(I) + cand.astype(np.uint64))
                ids = O.sample_user(cand, p, ns, ug)
                up = O.rng_uniform(self.seed, O.STREAM_POP_PICK, st, np.uint64(r) * np.uint64(I) + ids.astype(np.uint64))
                pl = idx.pop_idx[idx.pop_ptr[u]:idx.pop_ptr[u + 1]]
                xg, xp, kept = O.build_fake_pairs(ids, pl, up.astype(np.float32), idx.valid_item)
                rows += [r] * len(xg)
                gen += xg
                pop += xp
                base = int(idx.slot_ptr[u]) - s0       # the device writes the j-th sampled id to slot base+j
                pos += [base + j for j, kp in enumerate(kept) if kp]
            self.fake[b] = (np.asarray(rows, np.int64), np.asarray(gen, np.int64), np.asarray(pop, np.int64),
                            np.asarray(pos, np.int64))
        self.active = [b for b in range(self.n_batches) if len(self.fake[b][0]) > 0]
        self.order = np.arange(len(self.active))
        self.np_rng.shuffle(self.order)

    def _real(self, b):
        lo, hi = self._rows(b)
        r0, r1 = self.idx.real_ptr[lo], self.idx.real_ptr[hi]
        return self.idx.real_pop[r0:r1].astype(np.int64), self.idx.real_nic[r0:r1].astype(np.int64)

    def _slots(self, b):
        """the device's slot layout: fake pairs sit in per-user slot ranges (holes included), and the
        discriminator's dropout RNG is indexed by the logical row = slot position"""
        lo, hi = self._rows(b)
        s0, s1 = int(self.idx.slot_ptr[lo]), int(self.idx.slot_ptr[hi])
        return s0, s1

    def _fake_in_slots(self, b):
        """expand the compact fake list into the slot array the device sees (-1 holes)"""
        lo, hi = self._rows(b)
        idx = self.idx
        s0, s1 = self._slots(b)
        gen = np.full(s1 - s0, -1, np.int64)
        pop = np.full(s1 - s0, -1, np.int64)
        rows, g, p = self.fake[b][:3]
        # pairs of a user are written to the FIRST slots of that user's range in ascending item order,
        # dropped pairs leave holes at their sampled position -> reconstruct through slot_pos if given
        if len(self.fake[b]) == 4:
            pos = self.fake[b][3]
            gen[pos] = g
            pop[pos] = p
        else:
            cursor = {}
            for r, gi, pi in zip(rows, g, p):
                base = int(idx.slot_ptr[lo + r]) - s0
                k = cursor.get(r, 0)
                gen[base + k] = gi
                pop[base + k] = pi
                cursor[r] = k + 1
        return gen, pop

    # ------------------------------------------------------------------ phase D
    def d_phase(self):
        out = []
        hs = self.hs
        self.d_log = []
        for j in range(self.S):
            loss = None
            for k in self.order:
                b = self.active[k]
                st = self._step()
                rp, rn = self._real(b)
                fg, fp = self._fake_in_slots(b)
                nr, nf = len(rp), len(fg)
                dm = _d_masks(self.seed, st, nr + nf, hs[1:], 0.7)
                vf = fg >= 0
                Tr = O.d_tower(self.D, rp, rn, [m[:nr] for m in dm], 0.7)
                Tf = O.d_tower(self.D, np.where(vf, fp, 0), np.where(vf, fg, 0), [m[nr:] for m in dm], 0.7)
                loss = -np.log(Tr["y"]).sum() - (np.log(1 - Tf["y"]) * vf).sum()
                gr = O.d_tower_backward(self.D, Tr, [m[:nr] for m in dm], 0.7, -(1 - Tr["y"]))
                gf = O.d_tower_backward(self.D, Tf, [m[nr:] for m in dm], 0.7, Tf["y"] * vf)
                self.adam.apply(self.D, {k2: gr[k2] + gf[k2] for k2 in gr}, O.D_KEYS)
                self.d_log.append(loss)
            out.append(loss)
        return out

    # ------------------------------------------------------------------ phase G
    def g_phase(self, max_steps=None):
        """max_steps: stop after that many generator updates (a fixture that only needs the phase's first steps)"""
        out = []
        hs = self.hs
        self.g_log = []
        for j in range(self.S):
            last = None
            for k in self.order:
                if max_steps is not None and len(self.g_log) >= max_steps:
                    return out
                b = self.active[k]
                anneal = O.anneal_value(self.update_count)
                self.update_count += 1
                st, dst = self._step(), self._step()
                lo, hi = self._rows(b)
                B = hi - lo
                X = self._X(b)
                fg, fp = self._fake_in_slots(b)
                vf = fg >= 0
                dm = _d_masks(self.seed, dst, len(fg), hs[1:], 0.7)
                Tf = O.d_tower(self.D, np.where(vf, fp, 0), np.where(vf, fg, 0), dm, 0.7)
                sum_y = float((Tf["y"] * vf).sum())
                rows, gen, _ = self.fake[b][:3]
                eps = O.rng_normal(self.seed, O.STREAM_VAE_EPS, st, np.arange(B * O.Z_DIM, dtype=np.uint64).reshape(B, O.Z_DIM))
                losses, g, _ = O.g_loss_and_grads(self.P, X, self._mask(st, B), 0.75, eps, anneal, self.lam, rows, gen, len(rows), sum_y,
                                                  1.0, np.float64, self.quant)
                self.adam.apply(self.P, g, O.G_KEYS)
                last = (losses["g_loss"], losses["vae_loss"], losses["gan_loss"], anneal)
                self.g_log.append(last)
            out.append(last)
        return out

    # ------------------------------------------------------------------ validation (train.py:333-348)
    def validate(self, tr_csr, te_csr, rng_step=0):
        Xv = np.asarray(tr_csr.toarray(), np.float64)
        n = Xv.shape[0]
        idx = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(self.I) + np.arange(self.I, dtype=np.uint64)[None, :])
        mask = (O.rng_uniform(self.seed, O.STREAM_VAE_DROPOUT, rng_step, idx).astype(np.float32) < np.float32(0.75)).astype(np.float64)
        F = O.vae_forward(self.P, Xv, mask, 0.75, np.zeros((n, O.Z_DIM)), 0.0, 1.0, np.float64, self.quant)
        pred = F["logits"].copy()          # ranking by logits == ranking by softmax probabilities
        pred[Xv > 0] = -np.inf
        held = np.asarray(te_csr.toarray())
        return (float(np.mean(O.ndcg_binary_at_k(pred, held, 100))), float(np.mean(O.recall_at_k(pred, held, 20))),
                float(np.mean(O.recall_at_k(pred, held, 50))))


def _d_masks(seed, step, n, widths, keep):
    out = []
    for stream, w in zip((O.STREAM_D_DROP_A, O.STREAM_D_DROP_B, O.STREAM_D_DROP_C), widths):
        idx = np.arange(n * w, dtype=np.uint64).reshape(n, w)
        out.append((O.rng_uniform(seed, stream, step, idx).astype(np.float32) < np.float32(keep)).astype(np.float64))
    return out
