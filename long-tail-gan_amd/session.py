"""The five `sess.run` call sites of the reference's training loop, served by the HIP library.

The reference builds its loss/optimizer glue with TF ops (Codes/train.py:140-164) and then only ever
runs five fetch/feed combinations (train.py:200, :300, :326, :339 and test.py:146).  A maintainer who
keeps the reference's Python loop replaces

    train.py:131-164   (generated_tags ... g_trainer)   by   g = adversarial_graph(vae, generator_out, g_vae_loss, disc)
    train.py:169-172   tf.Session(...) / sess.run(init)   by   sess = Session(vae.engine)

and the loop body runs unchanged: the feeds are the reference's own (dense float32 X, python lists of
ids, the dense `generated_tags` mask, `sampled_cnt`, keep probabilities, anneal, gen_lambda).  The
facade converts them to the device formats of include/ltg.h (CSR rows, (row, id) pair lists) on
every call -- that conversion is host work the device-resident `ltgan.trainer.Trainer` does once per
dataset, so this is the compatibility path, not the fast one.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from .dataset import batch_csc
from .engine import CsrRows, Pairs
from .generator import Fetch, Placeholder


class AdversarialGraph:
    """Handles created by train.py:131-164 (same names)."""

    def __init__(self, vae, generator_out, g_vae_loss, disc):
        (self.y_data, self.y_generated, self.d_params, self.x_generated_id, self.x_popular_n_id, self.x_popular_g_id,
         self.x_niche_id, self.item_feature_arr, self.keep_prob) = disc
        self.vae, self.generator_out, self.g_vae_loss = vae, generator_out, g_vae_loss
        self.generated_tags = Placeholder("generated_tags")          # train.py:131
        self.sampled_cnt = Placeholder("sampled_cnt", 1.0)           # train.py:150
        self.gen_lambda = Placeholder("gen_lambda", 1.0)             # train.py:151
        self.d_loss_mean = Fetch("d_loss_mean")                      # train.py:142-143
        self.g_loss_mean = Fetch("g_loss_mean")                      # train.py:154-155
        self.gan_loss = Fetch("gan_loss")                            # train.py:156
        self.d_trainer = Fetch("d_trainer")                          # train.py:162 (shared Adam: Q5)
        self.g_trainer = Fetch("g_trainer")                          # train.py:163


def adversarial_graph(vae, generator_out, g_vae_loss, disc, learning_rate=None):
    """learning_rate: train.py:160 `tf.train.AdamOptimizer(LEARNING_RATE)` lives in the block this call replaces."""
    if learning_rate is not None:
        vae.engine.set_learning_rate(learning_rate)
    return AdversarialGraph(vae, generator_out, g_vae_loss, disc)


class Session:
    """`run(fetches, feed_dict)` for the fetch sets the reference uses; anything else raises."""

    def __init__(self, engine):
        self.eng = engine
        self.dev = engine.device
        self._acts = None
        self._step = 0

    # ------------------------------------------------------------------ feed conversion
    def _t(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(-1))).to(self.dev)

    def _rows(self, X, with_csc):
        X = sp.csr_matrix(X, dtype=np.float32)
        X.eliminate_zeros()
        X.sort_indices()
        if X.shape[1] != self.eng.I:
            raise ValueError("input_ph feed has %d columns, the generator was built for %d items" % (X.shape[1], self.eng.I))
        B = X.shape[0]
        kw = {}
        if X.nnz and not np.all(X.data == 1.0):
            kw["values"] = self._t(X.data, np.float32)
        if with_csc:
            slot, uptr, rowidx, pos = batch_csc(X, 0, B, self.eng.I)
            kw.update(slot=self._t(slot, np.int32), uptr=self._t(uptr, np.int32), rowidx=self._t(rowidx, np.int32),
                      csr_pos=self._t(pos, np.int32), n_unique=len(uptr) - 1, uitem=self._t(np.flatnonzero(slot >= 0), np.int32))
        batch = CsrRows(self._t(X.indptr, np.int32), self._t(X.indices, np.int32), 0, B, **kw)
        if self._acts is None or self._acts.rows < B:
            self._acts = self.eng.new_acts(B)
        return batch, B

    @staticmethod
    def _feed(feed, ph, required=True):
        if ph in feed:
            return feed[ph]
        if ph.default is not None:
            return ph.default
        if required:
            raise KeyError("feed_dict lacks %r" % ph)
        return None

    def _fake_pairs(self, feed, names, need_rows):
        gen = np.asarray(self._by_name(feed, names, "x_generated"), dtype=np.int64).reshape(-1)
        pop = np.asarray(self._by_name(feed, names, "x_popular_g"), dtype=np.int64).reshape(-1)
        if gen.shape != pop.shape:
            raise ValueError("x_generated and x_popular_g differ in length")
        rows = None
        if need_rows:
            # train.py:215-251: the mask row of a user holds exactly the kept sampled ids, which were appended to
            # x_generated in ascending order per user -> row-major nonzeros of the mask ARE the pair list
            M = sp.csr_matrix(np.asarray(self._by_name(feed, names, "generated_tags")))
            M.eliminate_zeros()
            M.sort_indices()
            if M.nnz != len(gen) or not np.array_equal(M.indices, gen):
                raise ValueError("generated_tags does not match x_generated (train.py:239-245 builds them together)")
            rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
        return Pairs(self._t(pop, np.int32), self._t(gen, np.int32), None if rows is None else self._t(rows, np.int32))

    @staticmethod
    def _by_name(feed, names, name):
        if name not in names:
            raise KeyError("feed_dict lacks placeholder %s" % name)
        return feed[names[name]]

    # ------------------------------------------------------------------ the five run signatures
    def run(self, fetches, feed_dict=None):
        feed = feed_dict or {}
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        want = [f.name for f in flist]
        names = {ph.name: ph for ph in feed}
        self._step += 1
        if "d_trainer" in want:
            out = self._run_d(want, feed, names)
        elif "g_trainer" in want:
            out = self._run_g(want, feed, names)
        elif want == ["generator_out"]:
            out = {"generator_out": self._run_forward(feed, names)}
        else:
            raise NotImplementedError("fetch set %s is not one the reference's loop runs (train.py:200,300,326,339)" % want)
        vals = [out.get(n) for n in want]
        return vals[0] if single else vals

    def _vae_feeds(self, feed, names):
        g = lambda n, d: float(feed[names[n]]) if n in names else d
        return g("keep_prob_ph", 0.75), g("is_training_ph", 0.0), g("anneal_ph", 1.0)     # MultiVAE.py:31,101,102

    def _run_forward(self, feed, names):
        batch, B = self._rows(self._by_name(feed, names, "input_ph"), with_csc=False)
        keep, is_tr, _ = self._vae_feeds(feed, names)
        probs = torch.empty(B, self.eng.I, dtype=torch.float32, device=self.dev)
        self.eng.forward(batch, self._acts, keep, is_tr, rng_step=2 * self._step, probs_out=probs)
        return probs.cpu().numpy()

    def _run_d(self, want, feed, names):
        real = Pairs(self._t(self._by_name(feed, names, "x_popular_n"), np.int32), self._t(self._by_name(feed, names, "x_niche"), np.int32))
        fake = self._fake_pairs(feed, names, need_rows=False)
        keep = float(self._by_name(feed, names, "keep_prob"))
        loss = self.eng.d_step(real, fake, keep, rng_step=2 * self._step)
        return {"d_loss_mean": np.float32(loss[0].item())}

    def _run_g(self, want, feed, names):
        batch, B = self._rows(self._by_name(feed, names, "input_ph"), with_csc=True)
        fake = self._fake_pairs(feed, names, need_rows=True)
        keep, is_tr, anneal = self._vae_feeds(feed, names)
        d_keep = float(self._by_name(feed, names, "keep_prob"))
        cnt = float(feed[names["sampled_cnt"]]) if "sampled_cnt" in names else 1.0
        lam = float(feed[names["gen_lambda"]]) if "gen_lambda" in names else 1.0
        if cnt != int(cnt) or cnt < 1:
            raise ValueError("sampled_cnt must be a positive count (train.py:247)")
        cnt_t = torch.tensor([int(cnt)], dtype=torch.int32, device=self.dev)
        # distinct counter streams for the generator's and the discriminator's dropout inside one step
        loss = self.eng.g_step(batch, fake, self._acts, cnt_t, anneal, lam, keep, is_tr, d_keep, rng_step=2 * self._step,
                               d_rng_step=2 * self._step + 1).cpu().numpy()
        return {"g_loss_mean": np.float32(loss[0]), "g_vae_loss": np.float32(loss[1]), "gan_loss": np.float32(loss[2])}
