"""The adversarial training loop of the reference (Codes/train.py:180-356) on device-resident data.

One global epoch = three phases (Q6):
  C  data creation: per batch one generator forward (dropout ON, eps OFF: train.py:200 feeds only
     input_ph) + fake-pair sampling (train.py:212-251)              -> Trainer.create_phase
  D  NUM_SUB_EPOCHS passes of discriminator updates over the cached batches (train.py:287-303)
  G  NUM_SUB_EPOCHS passes of generator updates (train.py:307-329), anneal = min(cap, n/20000)
then validation (train.py:333-348).  Batches whose sampling produced no valid fake pair are
skipped for the whole epoch (train.py:254-255); batch order is shuffled once per epoch (:284-285).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .dataset import DeviceData, EvalData
from .engine import Engine, Pipe


CREATE_LOGITS_BYTES = 1 << 30      # logits of one phase-C span (rows x I x 4)
PIPE_MAX_ITEMS = 1 << 30           # item slabs below this run the G step through ltg_g_step_sharded (one call, pipelined).  Was 65 536 while
                                   # the step's fork / join were event pairs (at 200 000 items the gain was within the noise); with the
                                   # device-word hand-over: phase G 279-281 -> 264 ms per epoch at 200 000 items, so every slab the call serves
TOWER_PAIRS = 1 << 17              # fake pairs per launch of the batched fake tower (activations: pairs x (h1 + h2 + 2 h3) x 4 bytes)


class Trainer:
    def __init__(self, engine: Engine, data: DeviceData, num_sub_epochs=10, gan_lambda=1.0, total_anneal_steps=20000,
                 anneal_cap=0.2, vae_keep=0.75, d_keep=0.7, shuffle_seed=0, span_create=None, batched_tower=None, pipe_step=None, step_log=0):
        """step_log = n: the losses of the FIRST n steps of every D phase and G phase are kept one row per step (d_step_log [n, 8]: d_loss;
        g_step_log [n, 8]: g_loss, vae_loss, gan_loss) -- what SURVEY 8/d6's per-step parity gate compares (tests/test_gpu_trajectory.py).  The
        per-sub-epoch rows d_losses / g_losses (the reference prints the sub-epoch's LAST step: train.py:300-303, :326-329) are unaffected."""
        self.eng, self.data = engine, data
        if span_create is None:
            span_create = os.environ.get("LTGAN_SPAN_CREATE", "1") != "0"    # measurement switch
        self.S = int(num_sub_epochs)
        self.lam = float(gan_lambda)
        self.total_anneal_steps, self.anneal_cap = total_anneal_steps, anneal_cap
        self.vae_keep, self.d_keep = vae_keep, d_keep
        self.update_count = 0.0                      # train.py:178
        self.rng_step = 0                            # every call gets a fresh RNG counter
        self.np_rng = np.random.RandomState(shuffle_seed)
        self.acts = engine.new_acts(data.max_rows)
        engine.workspace(data.max_rows, data.max_pairs)
        # phase C over spans of batches: as many as fit CREATE_LOGITS_BYTES of logits (1 = batch by batch)
        self.span_batches = 1
        # (measured, 64 batches: 1 000 items 5.5 -> 1.1 ms, 20 000 items 5.1 -> 2.6 ms; 200 000 items 8.7 -> 16.8 ms -- there the
        # streaming decoder kernel of a 100-row batch beats the generic one on 1 300 rows, so large slabs stay batch by batch)
        if span_create and data.n_batches > 1 and engine.I < 65536:
            self.span_batches = int(max(1, min(data.n_batches, CREATE_LOGITS_BYTES // (4 * engine.I * data.BS))))
        self.acts_c = engine.new_acts(min(data.N, self.span_batches * data.BS)) if self.span_batches > 1 else self.acts
        # phase G: the discriminator is fixed, so the fake tower of every G step of a sub-epoch is evaluated ahead in a few large
        # launches (Engine.fake_tower_batched) instead of three small ones inside every step
        self.batched_tower = os.environ.get("LTGAN_BATCHED_TOWER", "1") != "0" if batched_tower is None else bool(batched_tower)
        self.batched_tower = self.batched_tower and data.n_slots > 0
        if self.batched_tower:
            self.y_all = torch.zeros(self.S * data.n_slots, dtype=torch.float32, device=engine.device)       # [sub-epoch][slot]
            self._seg_step = torch.zeros(self.S, data.n_batches, dtype=torch.int64, device=engine.device)
            self._towers = data.tower_chunks(TOWER_PAIRS)
            engine.workspace(data.max_rows, max(data.max_pairs, max(t["fake"].n for t in self._towers)))
        self.active = list(range(data.n_batches))
        self.order = np.arange(data.n_batches)
        # large item slabs: the whole G step as ONE call with the decoder weight update and the lazy clock's slice running beside
        # the next step (ltg_g_step_sharded; here without a communicator).  LTGAN_PIPE_STEP: 0 = off, 1 = whenever the library
        # supports the configuration, default = slabs below PIPE_MAX_ITEMS (= all of them since the device-word hand-over)
        self.pipe, self.comm = None, None
        mode = os.environ.get("LTGAN_PIPE_STEP", "auto") if pipe_step is None else ("1" if pipe_step else "0")
        if mode != "0" and engine.sharded_step_ok(data.max_rows) and (mode == "1" or engine.I < PIPE_MAX_ITEMS):
            self.pipe = Pipe(engine, data.max_rows, 1, flags=int(os.environ.get("LTGAN_PIPE_FLAGS", "0")))
        dev = engine.device
        self.d_losses = torch.zeros(max(1, self.S), 8, dtype=torch.float32, device=dev)
        self.g_losses = torch.zeros(max(1, self.S), 8, dtype=torch.float32, device=dev)
        self.probe_hook = None                       # bench.py: (kind, batch) -> ltg_probe or None
        self.step_log = int(step_log)
        self.d_step_log = torch.zeros(max(1, self.step_log), 8, dtype=torch.float32, device=dev)
        self.g_step_log = torch.zeros(max(1, self.step_log), 8, dtype=torch.float32, device=dev)

    def _step(self):
        self.rng_step += 1
        return self.rng_step

    # ---------------------------------------------------------------- phase C (train.py:192-269)
    def create_phase(self):
        d, eng = self.data, self.eng
        d.fake_cnt.zero_()
        if self.span_batches > 1:
            # no weight moves in this phase, so consecutive batches go through ONE forward and ONE sampler launch; every batch
            # keeps its own RNG counter and local row numbers (ltg_fwd_opts.rows_per_step): the same draws as batch by batch
            for b0 in range(0, d.n_batches, self.span_batches):
                b1 = min(d.n_batches, b0 + self.span_batches)
                v = d.span(b0, b1)
                st = self._step()
                self.rng_step += b1 - b0 - 1                         # batch b0 + k: counter st + k
                eng.forward(v["batch"], self.acts_c, keep_prob=self.vae_keep, is_training=0.0, rng_step=st, rows_per_step=d.BS)
                v["samp"].rng_step = st
                eng.sample_pairs(v["samp"], self.acts_c, d.fake_gen, d.fake_pop, d.fake_cnt[b0:])
        else:
            for b in range(d.n_batches):
                v = d.view(b)
                st = self._step()
                eng.forward(v["batch"], self.acts, keep_prob=self.vae_keep, is_training=0.0, rng_step=st)
                v["samp"].rng_step = st
                eng.sample_pairs(v["samp"], self.acts, d.fake_gen, d.fake_pop, d.fake_cnt[b:])
        cnt = d.fake_cnt.cpu().numpy()               # the only host sync of the phase
        self.active = [b for b in range(d.n_batches) if cnt[b] > 0]      # train.py:254-255
        self.order = np.arange(len(self.active))
        self.np_rng.shuffle(self.order)              # train.py:284-285
        user_err_cnt = int((~d.idx.user_ok).sum())
        return user_err_cnt

    # ---------------------------------------------------------------- phase D (train.py:287-303)
    def d_phase(self):
        d, eng = self.data, self.eng
        eng.pin_stream()
        try:
            i = 0
            for j in range(self.S):
                for n, k in enumerate(self.order):
                    v = d.view(self.active[k])
                    pr = self.probe_hook("d", self.active[k]) if self.probe_hook else None
                    logged = i < self.step_log                   # (a logged step that ends its sub-epoch is copied into the sub-epoch's row)
                    eng.d_step(v["real"], v["fake"], keep_prob=self.d_keep, rng_step=self._step(),
                               loss_out=self.d_step_log[i] if logged else self.d_losses[j], probe=pr)
                    if logged and n == len(self.order) - 1:
                        self.d_losses[j].copy_(self.d_step_log[i])
                    i += 1
        finally:
            eng.pin_stream(False)           # (also when a step raised: nothing may stay pinned to a stale stream handle)
        if eng._dfork is not None and eng.check_on_flush:
            eng.check_pipes()               # (the D steps' fork: a device-side wait that gave up poisons it -- one host sync per phase)
        return self.d_losses

    # ---------------------------------------------------------------- phase G (train.py:307-329)
    def anneal(self):
        if self.total_anneal_steps > 0:
            return min(self.anneal_cap, 1.0 * self.update_count / self.total_anneal_steps)
        return self.anneal_cap

    def _tower_ahead(self):
        """y_generated of every G step of the coming phase.  The k-th step of sub-epoch j will draw the tower's dropout with the
        counter rng_step + 2 (j n_active + k) + 2 (every step takes two counters: generator, then tower)."""
        if not self.batched_tower:
            return
        nb, na = self.data.n_batches, len(self.order)
        steps = np.zeros((self.S, nb), np.int64)
        for j in range(self.S):
            for k, pos in enumerate(self.order):
                steps[j, self.active[pos]] = self.rng_step + 2 * (j * na + k) + 2
        self._tower_steps = steps                              # (host copy: every G step checks the counter it really uses)
        self._seg_step.copy_(torch.from_numpy(steps))          # the one host -> device copy of the phase
        for j in range(self.S):
            for t in self._towers:
                self.eng.fake_tower_batched(t["fake"], t["seg_of"], t["seg_row0"], self._seg_step[j], self.y_all, self.d_keep,
                                            seg_off=t["seg_off"], y_off=j * self.data.n_slots + t["s0"])

    def check_pipe(self, sync=True):
        """raises if a device-side wait of the one-call G step's hand-overs gave up (Engine.check_pipes; every G phase ends with it)"""
        if self.pipe is not None:
            self.eng.check_pipes()

    def _g_begin(self):
        self.last_anneal = []
        self.eng.q0_defer = True        # lazy Adam clock of W_q0: one flush at the end of the phase
        self.eng.pin_stream()

    def _g_end(self, ok):
        """leaves the engine usable whatever happened inside the phase (a raised LtgError, a failed collective, the RNG-counter check)"""
        eng = self.eng
        try:
            if self.pipe is not None:
                eng.pipe_join(self.pipe)    # the forked weight update and clock slice of the last step
            eng.q0_defer = False
            if ok:
                dirty = eng._q0_dirty
                eng.g_flush()               # (+ Engine.check_pipes: one host sync per phase; raises if a hand-over wait gave up)
                if not dirty and eng.check_on_flush:
                    self.check_pipe()       # (a phase without a dirty clock still checks)
        finally:
            eng.q0_defer = False
            eng.pin_stream(False)

    def _g_one(self, j, b, v, a, loss_out=None):
        """one generator update (train.py:326) of batch b in sub-epoch j"""
        d, eng = self.data, self.eng
        loss_out = self.g_losses[j] if loss_out is None else loss_out
        rs, ds = self._step(), self._step()
        if self.batched_tower and ds != self._tower_steps[j, b]:
            raise RuntimeError("the fake tower of this step was evaluated ahead with another RNG counter")
        pr = self.probe_hook("g", b) if self.probe_hook else None
        if self.pipe is not None:
            go = eng.g_opts(d.fake_cnt[b:], a, self.lam, self.vae_keep, 1.0, self.d_keep, rs, ds, probe=pr,
                            y_pre=self.y_all if self.batched_tower else None, y_off=j * d.n_slots + v["slot0"])
            eng.g_step_sharded(v["batch"], v["fake"], self.acts, go, self.pipe, self.comm, loss_out=loss_out,
                               next_batch=getattr(self, "_next_batch", None))
        else:
            eng.g_step(v["batch"], v["fake"], self.acts, d.fake_cnt[b:], anneal=a, gan_lambda=self.lam,
                       keep_prob=self.vae_keep, is_training=1.0, d_keep_prob=self.d_keep, rng_step=rs,
                       d_rng_step=ds, loss_out=loss_out, probe=pr,
                       y_pre=self.y_all if self.batched_tower else None, y_off=j * d.n_slots + v["slot0"])

    def g_phase(self):
        d = self.data
        self._g_begin()
        ok = False
        try:
            self._tower_ahead()        # every fake tower of the phase in a few large launches
            seq = [self.active[k] for k in self.order]
            i = 0
            for j in range(self.S):
                a = self.anneal()
                for n, b in enumerate(seq):
                    a = self.anneal()
                    self.update_count += 1
                    # the batch of the NEXT step (known: the phase's order is fixed): its rows of W_q0 are caught up during this one
                    nb = seq[n + 1] if n + 1 < len(seq) else (seq[0] if j + 1 < self.S else None)
                    self._next_batch = d.view(nb)["batch"] if nb is not None else None
                    logged = i < self.step_log
                    self._g_one(j, b, d.view(b), a, loss_out=self.g_step_log[i] if logged else None)
                    if logged and n == len(seq) - 1:
                        self.g_losses[j].copy_(self.g_step_log[i])
                    i += 1
                self.last_anneal.append(a)
            ok = True
        finally:
            self._g_end(ok)
        return self.g_losses

    def epoch(self):
        """one global epoch's three phases; returns per-phase wall seconds (device-synchronised)."""
        t = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        err = self.create_phase()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.d_phase()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self.g_phase()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return dict(user_err_cnt=err, t_create=t1 - t0, t_d=t2 - t1, t_g=t3 - t2, t_total=t3 - t0)


EVAL_LOGITS_BYTES = 2 << 30


def eval_chunk_rows(n_items_local, budget=EVAL_LOGITS_BYTES):
    """users per scoring chunk so that the [chunk, I] fp32 logits stay within `budget` bytes (test.py:76 scores 20 000
    users at a time: 16 GB at I = 200 000)."""
    return max(1, int(budget // (4 * max(1, n_items_local))))


class Evaluator:
    """Validation / test scoring (train.py:333-348, test.py:138-173): forward with dropout ON (Q3),
    fold-in items masked to -inf, NDCG@100 / Recall@20 / Recall@50, in chunks of `chunk` users
    (test.py:76 uses 20000), capped so that a chunk's logits stay within EVAL_LOGITS_BYTES."""

    def __init__(self, engine: Engine, ev: EvalData, chunk=20000):
        self.eng, self.ev, self.chunk = engine, ev, int(min(chunk, max(1, ev.n), eval_chunk_rows(engine.I)))
        self.acts = engine.new_acts(self.chunk)
        self.out = torch.zeros(ev.n, 4, dtype=torch.float32, device=engine.device)

    def run(self, rng_step=0, keep_prob=0.75):
        eng, ev = self.eng, self.ev
        for lo in range(0, ev.n, self.chunk):
            hi = min(ev.n, lo + self.chunk)
            tr, te = ev.rows(lo, hi)
            eng.forward(tr, self.acts, keep_prob=keep_prob, is_training=0.0, rng_step=rng_step + lo)
            eng.rank_metrics(self.acts, tr, te, self.out[lo:])
        o = self.out.cpu().numpy().astype(np.float64)
        ok = o[:, 3] > 0
        n = int(ok.sum())
        return dict(ndcg=float(o[ok, 0].mean()) if n else float("nan"), recall20=float(o[ok, 1].mean()) if n else float("nan"),
                    recall50=float(o[ok, 2].mean()) if n else float("nan"), n_users=n)
