"""Item-sharded multi-GPU training: one process per GPU, rank r owns the item slab
[item_lo_r, item_hi_r) of W_q0 / W_p1t / b_p1 and their Adam moments plus the matching columns of the
interaction matrix; the middle layers and the whole discriminator are replicated (identical inputs
and identical counter-RNG streams give identical values on every rank, no broadcast needed).

Exchange steps per generator step (torch.distributed over RCCL/xGMI; the C ABI never communicates):
  1. all-reduce(sum)  encoder pre-activation   [B, H]      (240 KB at defaults)
  2. all-gather       row partials             [B, 5]      (max, sum exp, sum x*logit, sum_S exp, sum x)
  3. all-reduce(sum)  dh2                      [B, H]      -- issued asynchronously; the Adam update of the local W_p1t / b_p1
                                                              rows (ltg_g_bwd_dec1, the largest kernel of the step) runs under it
Phase C adds one all-reduce of the candidates' logits (~12 KB).
The sharded tables never leave their GPU and need no gradient all-reduce.

Discriminator step (train.py:300), two modes:
  replicated   every rank runs the whole step (identical inputs + identical counter RNG => identical weights, no exchange);
  pair split    rank r runs forward + backward over rows [r n / R, (r+1) n / R) of the real | fake pair batch
               (ltg_d_grad: dropout keyed by the GLOBAL row), ONE all-reduce(sum) of the gradient vector
               (P_D + 1 floats: 644 KB at defaults, 14.2 MB for the wide discriminator), then the identical Adam sweep on
               every rank (ltg_d_apply).
`d_split=None` picks the split when the discriminator has >= D_SPLIT_MIN_PARAMS trainable parameters: at config.ini's
sizes the whole replicated step is ~60 us of latency-bound launches on one GPU, less than the split's four launches plus a
latency-bound 644-KB all-reduce would take; at the wide sizes (3.5 M parameters, MFMA-bound) the split divides the work.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .engine import Pipe, _ptr
from .trainer import Trainer, eval_chunk_rows

D_SPLIT_MIN_PARAMS = 1_000_000


def item_slab(n_items, rank, world):
    """contiguous slab boundaries, multiples of 64 items except the last"""
    per = -(-n_items // world)
    per = -(-per // 64) * 64
    if (world - 1) * per >= n_items:
        raise ValueError("%d items cannot be cut into %d slabs of 64-item multiples (the last rank would own nothing): "
                         "use at most %d ranks" % (n_items, world, -(-n_items // 64)))
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


class ShardedTrainer(Trainer):
    def __init__(self, engine, data, group=None, d_split=None, transport=None, **kw):
        """transport of the one-call step's three exchanges: None = environment LTGAN_COMM, default "rccl" (RCCL bound directly on the nccl backend,
        host callbacks over the group otherwise); "host-ordered" = host callbacks adding in rank order; "oneshot" = the library's one-shot exchange
        over HIP-IPC-mapped staging buffers (_rccl.OneShotComm; correctness only)"""
        super().__init__(engine, data, **kw)
        self.transport = transport
        self.group = group
        self.R = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        dev = engine.device
        e = engine
        self.d_params = e.h0 * e.h1 + e.h1 + e.h0 * e.h2 + e.h2 + (e.h1 + e.h2) * e.h3 + e.h3 + e.h3 + 1
        self.d_split = (self.d_params >= D_SPLIT_MIN_PARAMS) if d_split is None else bool(d_split)
        self.d_grad = torch.zeros(engine.d_grad_floats(), dtype=torch.float32, device=dev)
        B = data.max_rows
        Bc = max(B, self.acts_c.rows)                  # rows of a phase-C span of batches
        self.rowpart = torch.zeros(Bc * 5, dtype=torch.float32, device=dev)
        self.rowpart_all = torch.zeros(self.R * Bc * 5, dtype=torch.float32, device=dev)
        self.dh2 = torch.zeros(B, engine.H, dtype=torch.float32, device=dev)
        self.fake_overlap = os.environ.get("LTGAN_FAKE_OVERLAP", "1") != "0"     # measurement switch
        self._side = torch.cuda.Stream(dev)
        self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        self.dec1_overlap = os.environ.get("LTGAN_DEC1_OVERLAP", "1") != "0"     # measurement switch
        self._ev_dlog, self._ev_dec1 = torch.cuda.Event(), torch.cuda.Event()
        engine.workspace(Bc, data.max_pairs)       # sized once: the side stream must never see it reallocated
        self._init_sharded_step(engine, B)
        self.cand_logit = torch.zeros(max(1, int(data.idx.cand_ptr[-1])), dtype=torch.float32, device=dev)

    def _init_sharded_step(self, engine, B):
        """ltg_g_step_sharded (one call per G step, exchanges in-stream) when the library serves this configuration.  Transport:
        RCCL bound directly (backend "nccl": the library calls ncclAllReduce / ncclAllGather itself on the step's stream), or host
        callbacks over the group for test rigs whose ranks share a GPU (gloo).  LTGAN_SHARDED_STEP=0: the cut-point sequence."""
        from ._rccl import HostComm, OneShotComm, RcclComm
        self.pipe, self.comm = None, None
        mine = os.environ.get("LTGAN_SHARDED_STEP", "1") != "0" and engine.sharded_step_ok(B)
        # every rank must take the same path (a rank whose slab the one-call step does not serve -- e.g. a last slab that is not a
        # multiple of 8 items -- would otherwise wait in torch.distributed collectives the others never issue)
        agree = torch.tensor([1 if mine else 0], dtype=torch.int32, device=engine.device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=self.group)
        if int(agree.item()) == 0:
            return
        self.pipe = Pipe(engine, B, self.R, flags=int(os.environ.get("LTGAN_PIPE_FLAGS", "0")))
        transport = self.transport or os.environ.get("LTGAN_COMM", "rccl")
        if transport == "oneshot":
            # the library's own one-shot exchange over HIP-IPC-mapped staging buffers (a second transport, correctness only: _rccl.OneShotComm)
            self.comm = OneShotComm(self.group, engine.device, max(self.pipe.h1pre.numel(), self.pipe.rowpart_all.numel() // self.R),
                                    limit_ms=int(os.environ.get("LTGAN_ONESHOT_LIMIT_MS", "0")))
            return
        if transport == "host-ordered":
            self.comm = HostComm(self.group, self.pipe.buffers(), ordered=True)
            return
        if dist.get_backend(self.group) == "nccl" and transport == "rccl":
            comm, err = None, ""
            try:
                comm = RcclComm(self.group, engine.device, warm_counts=(self.pipe.h1pre.numel(), self.pipe.rowpart_all.numel() // self.R))
            except Exception as e:                # (every rank must take the same branch: agree below)
                err = repr(e)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=engine.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 1:
                self.comm = comm
                return
            if comm is not None:
                comm.close()
            if self.rank == 0:
                print("ltgan.sharded: direct RCCL communicator unavailable (%s); exchanges go through torch.distributed" % err, flush=True)
        self.comm = HostComm(self.group, self.pipe.buffers())

    def close(self):
        """destroys the step's own RCCL communicator (ncclCommDestroy); call before torch.distributed.destroy_process_group()"""
        if self.pipe is not None:
            self.eng.pipe_join(self.pipe)
            torch.cuda.synchronize(self.eng.device)
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def abort(self):
        """the exception path: give up the communicator without waiting for what is in flight on it (ncclCommAbort), no device synchronisation"""
        if self.comm is not None:
            self.comm.abort()
            self.comm = None

    # `with ShardedTrainer(...) as tr:` -- close() on the way out, abort() when an exception is passing through (a blocking destroy on ONE
    # rank's error path would leave that rank in ncclCommDestroy and its peers in an in-stream collective until RCCL's timeout)
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:
            self.abort()
        return False

    # -- collectives -------------------------------------------------------------------------------
    def _allreduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _allgather_rowpart(self, B):
        n = B * 5
        out = self.rowpart_all[: self.R * n]
        src = self.rowpart[:n]
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, src, group=self.group)
        else:
            parts = [out[r * n:(r + 1) * n] for r in range(self.R)]
            dist.all_gather(parts, src, group=self.group)
        return out

    # -- forward over the shards ---------------------------------------------------------------------
    def _forward(self, v, fake, fopts, acts=None):
        acts = self.acts if acts is None else acts
        eng, B = self.eng, v["batch"].n_rows
        eng.g_fwd_enc(v["batch"], acts, fopts)
        self._allreduce(acts.h1[:B])
        eng.g_fwd_rest(v["batch"], fake, acts, fopts, self.rowpart)
        return self._allgather_rowpart(B)

    def create_phase(self):
        d, eng = self.data, self.eng
        d.fake_cnt.zero_()
        # spans of batches (Trainer.create_phase): three collectives per SPAN instead of per batch
        nb = self.span_batches
        for b0 in range(0, d.n_batches, nb):
            b1 = min(d.n_batches, b0 + nb)
            v = d.span(b0, b1) if nb > 1 else d.view(b0)
            acts = self.acts_c if nb > 1 else self.acts
            st = self._step()
            self.rng_step += b1 - b0 - 1
            B = v["batch"].n_rows
            fo = eng.fwd_opts(self.vae_keep, 0.0, st)
            fo.rows_per_step = d.BS if nb > 1 else 0
            rp_all = self._forward(v, None, fo, acts)
            eng.rowstats_combine(rp_all, self.R, B, acts.lse)
            c0, c1 = int(d.idx.cand_ptr[v["lo"]]), int(d.idx.cand_ptr[v["hi"]])
            eng.gather_cand_logits(v["samp"], acts, self.cand_logit)
            if c1 > c0:
                self._allreduce(self.cand_logit[c0:c1])
            v["samp"].rng_step = st
            v["samp"].cand_logit = _ptr(self.cand_logit)
            eng.sample_pairs(v["samp"], acts, d.fake_gen, d.fake_pop, d.fake_cnt[b0:])
        cnt = d.fake_cnt.cpu().numpy()
        self.active = [b for b in range(d.n_batches) if cnt[b] > 0]
        self.order = np.arange(len(self.active))
        self.np_rng.shuffle(self.order)          # same seed on every rank -> same order
        return int((~d.idx.user_ok).sum())

    # -- discriminator phase ---------------------------------------------------------------------------
    def d_phase(self):
        if not self.d_split or self.R == 1:
            return super().d_phase()          # replicated: identical on every rank
        d, eng = self.data, self.eng
        for j in range(self.S):
            for k in self.order:
                v = d.view(self.active[k])
                n = v["n_real"] + v["n_slots"]
                lo, hi = self.rank * n // self.R, (self.rank + 1) * n // self.R
                eng.d_grad(v["real"], v["fake"], lo, hi, self.d_grad, keep_prob=self.d_keep, rng_step=self._step())
                self._allreduce(self.d_grad)
                eng.d_apply(self.d_grad, loss_out=self.d_losses[j])
        return self.d_losses

    def g_phase(self):
        out = super().g_phase()
        # the one-shot transport's waits are bounded: a peer's message that never came leaves garbage behind a counter, not a hang -- the phase
        # ends by looking at it (RCCL and the host transports raise through their own error paths)
        waits = getattr(self.comm, "expired_waits", None)
        if waits is not None:
            n = waits()
            if n > 0:
                from ._cabi import LtgError
                raise LtgError("one-shot exchange: %d device-side waits for a peer's message gave up during the phase; the model is not to be trusted" % n)
        return out

    def _g_one(self, j, b, v, a, loss_out=None):
        """one generator update of batch b over the item shards (loss_out: Trainer.step_log's per-step row)"""
        d, eng = self.data, self.eng
        loss_out = self.g_losses[j] if loss_out is None else loss_out
        B = v["batch"].n_rows
        pr = self.probe_hook("g", b) if self.probe_hook else None
        rs, ds = self._step(), self._step()
        if self.batched_tower and ds != self._tower_steps[j, b]:
            raise RuntimeError("the fake tower of this step was evaluated ahead with another RNG counter")
        go = eng.g_opts(d.fake_cnt[b:], a, self.lam, self.vae_keep, 1.0, self.d_keep, rs, ds, probe=pr,
                        y_pre=self.y_all if self.batched_tower else None, y_off=j * d.n_slots + v["slot0"])
        if self.pipe is not None:
            # ONE call: every launch of the step and its three exchanges in-stream (ltg_g_step_sharded)
            eng.g_step_sharded(v["batch"], v["fake"], self.acts, go, self.pipe, self.comm, loss_out=loss_out,
                               next_batch=getattr(self, "_next_batch", None))
            return
        # the step cut at its exchange points, collectives through torch.distributed (configurations ltg_g_step_sharded does not
        # serve: fp32 decoder operands, small slabs, the dense W_q0 sweep)
        if self.fake_overlap and v["fake"].n > 0 and not self.batched_tower:
            # the fake tower (replicated, needs nothing of the generator) on a side stream, beside the forward and its
            # two exchanges; ordered after the previous step's reader of its outputs, joined before this step's
            main = torch.cuda.current_stream()
            self._ev_fork.record(main)
            self._side.wait_event(self._ev_fork)
            eng.g_fake_tower(v["batch"], v["fake"], go, stream=self._side)
            self._ev_join.record(self._side)
        rp_all = self._forward(v, v["fake"], go.fwd)
        if go.fake_done:
            torch.cuda.current_stream().wait_event(self._ev_join)
        eng.g_bwd_dec(v["batch"], v["fake"], self.acts, go, rp_all, self.R, loss_out, self.dh2)
        if self.dec1_overlap:
            # the decoder weight update (HBM-bound, the largest kernel; needs only dlog and h2) on the side stream with 224
            # of its 256 workgroups; exchange 3 and then the rest of the backward chain (which needs only the all-reduced
            # dh2) run beside it on the CUs that leaves free.  Joined before the next forward reads W_p1t.
            main = torch.cuda.current_stream()
            self._ev_dlog.record(main)
            self._side.wait_event(self._ev_dlog)
            eng.g_bwd_dec1(v["batch"], v["fake"], self.acts, go, stream=self._side)
            self._ev_dec1.record(self._side)
            self._allreduce(self.dh2[:B])
            eng.g_bwd_rest(v["batch"], v["fake"], self.acts, go, self.dh2)
            main.wait_event(self._ev_dec1)
        else:
            # exchange 3 flies while the decoder weight update (which needs none of it) runs
            work = dist.all_reduce(self.dh2[:B], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            eng.g_bwd_dec1(v["batch"], v["fake"], self.acts, go)
            work.wait()
            eng.g_bwd_rest(v["batch"], v["fake"], self.acts, go, self.dh2)


class ShardedEvaluator:
    """Validation / test scoring over item shards (train.py:333-348, test.py:138-173).  Per chunk of users: the
    sharded forward (one all-reduce of the encoder pre-activation), then two small exchanges for the ranking:
      1. all-reduce(sum) of the held-out entries' scores   (float32 per held-out entry; the owner contributes)
      2. all-reduce(sum) of the per-entry rank counts      (int32 per held-out entry)
    The softmax is never materialised: ranking by logits equals ranking by probabilities row by row.  Every rank
    ends with the identical metric table."""

    def __init__(self, engine, ev, group=None, chunk=20000):
        self.eng, self.ev, self.group = engine, ev, group
        self.chunk = int(min(chunk, max(1, ev.n), eval_chunk_rows(engine.I)))
        self.acts = engine.new_acts(self.chunk)
        dev = engine.device
        n_te = max(1, int(ev.te_indices.numel()))
        self.score = torch.zeros(n_te, dtype=torch.float32, device=dev)
        self.count = torch.zeros(n_te, dtype=torch.int32, device=dev)
        self.out = torch.zeros(ev.n, 4, dtype=torch.float32, device=dev)
        self.rowpart = torch.zeros(self.chunk * 5, dtype=torch.float32, device=dev)

    def run(self, rng_step=0, keep_prob=0.75):
        eng, ev = self.eng, self.ev
        te_ptr = ev.te_host.indptr
        for lo in range(0, ev.n, self.chunk):
            hi = min(ev.n, lo + self.chunk)
            tr, te = ev.rows(lo, hi)
            fo = eng.fwd_opts(keep_prob, 0.0, rng_step + lo)
            eng.g_fwd_enc(tr, self.acts, fo)
            dist.all_reduce(self.acts.h1[: hi - lo], op=dist.ReduceOp.SUM, group=self.group)
            eng.g_fwd_rest(tr, None, self.acts, fo, self.rowpart)
            e0, e1 = int(te_ptr[lo]), int(te_ptr[hi])
            eng.rank_scores(self.acts, tr, te, self.score)
            if e1 > e0:
                dist.all_reduce(self.score[e0:e1], op=dist.ReduceOp.SUM, group=self.group)
            eng.rank_counts(self.acts, tr, te, self.score, self.count)
            if e1 > e0:
                dist.all_reduce(self.count[e0:e1], op=dist.ReduceOp.SUM, group=self.group)
            eng.rank_finish(te, self.count, self.out[lo:])
        o = self.out.cpu().numpy().astype(np.float64)
        ok = o[:, 3] > 0
        n = int(ok.sum())
        return dict(ndcg=float(o[ok, 0].mean()) if n else float("nan"), recall20=float(o[ok, 1].mean()) if n else float("nan"),
                    recall50=float(o[ok, 2].mean()) if n else float("nan"), n_users=n)
