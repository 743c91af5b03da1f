"""Host-side engine: owns every device buffer (torch is only the allocator / stream provider) and
drives the HIP kernels through the C ABI of include/ltg.h.

One Engine == the state the reference keeps inside its TF session (Codes/train.py:125-174):
generator variables (MultiVAE.py:188-230), discriminator variables (discriminator.py:14-41), ONE
AdamOptimizer whose step counter is shared by the D and the G updates (train.py:160-164, Q5).
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np
import torch

from . import _cabi as cabi

G_NAMES = ["weight_q_0to1", "weight_q_1to2", "weight_p_0to1", "weight_p_1to2",
           "bias_q_1", "bias_q_2", "bias_p_1", "bias_p_2"]          # MultiVAE.py:196-197,216-217
D_NAMES = ["d_w1_1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3", "d_w4", "d_b4"]  # discriminator.py:23-41


def _ptr(t, off=0):
    if t is None:
        return None
    return t.data_ptr() + off * t.element_size()


def _pp(probe):
    return C.pointer(probe) if probe is not None else None


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise cabi.LtgError("no HIP device visible: the Long-Tail-GAN path has no CPU fallback")
    return torch.device(device)


def init_generator_host(I, H, Z, seed):
    """Default initialisers of MultiVAE._construct_weights (MultiVAE.py:188-230) in engine layout
    [W_q0 [I,H], W_q1 [H,2Z], W_p0 [Z,H], W_p1t [I,H], b_q0 [H], b_q1 [2Z], b_p0 [H], b_p1 [I]]: weights
    tf.contrib.layers.xavier_initializer = uniform(+-sqrt(6 / (fan_in + fan_out))) (:199-202, :219-221), biases
    tf.truncated_normal_initializer(stddev=0.001) = N(0, 0.001) re-drawn beyond 2 sigma (:204-207, :223-225).  The
    random STREAM is torch's (TF's cannot be reproduced without TF); the distributions are the reference's."""
    g = torch.Generator().manual_seed(seed)

    def xavier(fi, fo, shape):
        lim = math.sqrt(6.0 / (fi + fo))
        return (torch.rand(shape, generator=g) * 2 - 1) * lim

    def tn(shape, std):
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
        return t

    host = [xavier(I, H, (I, H)), xavier(H, 2 * Z, (H, 2 * Z)), xavier(Z, H, (Z, H)),
            xavier(H, I, (I, H)),                                # W_p1 [H, I] stored item-major [I][H]
            tn((H,), 1e-3), tn((2 * Z,), 1e-3), tn((H,), 1e-3), tn((I,), 1e-3)]
    return [t.numpy() for t in host]


def init_discriminator_host(feature_len, h_sizes, seed):
    """Default initialisers of discriminator.py:14-41: every matrix tf.truncated_normal(stddev=0.1) (the frozen embedding
    :14, w1 :23, w2 :28, w3 :36, w4 :40), every bias tf.zeros (:24, :29, :37, :41).
    -> (emb [F,h0], [w1 [h0,h1], b1, w2 [h0,h2], b2, w3 [h1+h2,h3], b3, w4 [h3], b4 [1]])."""
    g = torch.Generator().manual_seed(seed)

    def tn(shape):
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, 0.0, 0.1, -0.2, 0.2, generator=g)
        return t

    h0, h1, h2, h3 = h_sizes
    emb = tn((feature_len, h0))
    host = [tn((h0, h1)), torch.zeros(h1), tn((h0, h2)), torch.zeros(h2), tn((h1 + h2, h3)), torch.zeros(h3), tn((h3,)), torch.zeros(1)]
    return emb.numpy(), [t.numpy() for t in host]


class Acts:
    """Caller-owned activations of one generator forward (ltg_gen_acts)."""

    def __init__(self, rows, n_items, H, Z, device):
        f = dict(dtype=torch.float32, device=device)
        self.rows = rows
        self.h1 = torch.empty(rows, H, **f)
        self.mulv = torch.empty(rows, 2 * Z, **f)
        self.z = torch.empty(rows, Z, **f)
        self.h2 = torch.empty(rows, H, **f)
        self.logits = torch.empty(rows, n_items, **f)
        self.lse = torch.empty(rows, **f)
        self.kl_rows = torch.empty(rows, **f)
        self.row_scale = torch.empty(rows, **f)
        self.c = cabi.ltg_gen_acts(_ptr(self.h1), _ptr(self.mulv), _ptr(self.z), _ptr(self.h2), _ptr(self.logits),
                                   _ptr(self.lse), _ptr(self.kl_rows), _ptr(self.row_scale))


class CsrRows:
    """A range of user rows of a device-resident CSR matrix (ltg_batch), optionally with the transposed
    view (entries grouped by item) that the G step needs."""

    def __init__(self, indptr, indices, row_lo, row_hi, values=None, slot=None, uptr=None, rowidx=None, csr_pos=None,
                 n_unique=0, slot_off=0, uptr_off=0, ent_off=0, row_norm2=None, uitem=None, uitem_off=None):
        self.keep = (indptr, indices, values, slot, uptr, rowidx, csr_pos, row_norm2, uitem)
        self.n_rows = int(row_hi - row_lo)
        self.c = cabi.ltg_batch(self.n_rows, int(n_unique), _ptr(indptr, row_lo), _ptr(indices), _ptr(values),
                                _ptr(slot, slot_off), _ptr(uptr, uptr_off), _ptr(rowidx, ent_off), _ptr(csr_pos, ent_off),
                                _ptr(row_norm2, row_lo), _ptr(uitem, uptr_off if uitem_off is None else uitem_off))


class Pairs:
    def __init__(self, pop, niche, row=None, n=None, off=0):
        self.keep = (pop, niche, row)
        self.n = int(pop.numel() - off if n is None else n)
        self.c = cabi.ltg_pairs(self.n, 0, _ptr(pop, off), _ptr(niche, off), _ptr(row, off))


class Pipe:
    """Caller-owned side stream, events and exchange buffers of ltg_g_step_sharded (include/ltg.h: ltg_pipe) for batches of up to
    `rows` rows exchanged between `n_ranks` ranks."""

    def __init__(self, engine, rows, n_ranks=1, flags=0):
        from ._hip import EventPair
        dev = engine.device
        self.side_stream = torch.cuda.Stream(dev)
        # the Adam tail's own stream: only with LTG_PIPE_TAIL_OWN (device-word hand-over; the default keeps the tail on the caller's stream)
        self.tail_stream = torch.cuda.Stream(dev) if (int(flags) & cabi.LTG_PIPE_TAIL_OWN) else None
        self._ev = (EventPair(timing=False), EventPair(timing=False))     # (fork, dec1), (tail, -)
        f = dict(dtype=torch.float32, device=dev)
        self.h1pre = torch.zeros(rows, engine.H, **f)
        self.rowpart_all = torch.zeros(n_ranks * rows * 5, **f)
        self.dh2 = torch.zeros(rows, engine.H, **f)
        self.sync = torch.zeros(16, dtype=torch.int32, device=dev)        # words of the device-side hand-overs; [2] = waits that gave up = the pipe's poison
        # catch-up ahead (include/ltg.h, ABI v13): "the current batch holds this row" marks; LTGAN_Q0_AHEAD=0: every call launches its own catch-up
        self.q0_mark = torch.zeros(engine.I, dtype=torch.int32, device=dev) if os.environ.get("LTGAN_Q0_AHEAD", "1") != "0" else None
        self.c = cabi.ltg_pipe(self.side_stream.cuda_stream, self._ev[0].start, self._ev[0].stop, self._ev[1].start, _ptr(self.h1pre),
                               _ptr(self.rowpart_all), _ptr(self.dh2), int(flags), 0, _ptr(self.sync),
                               self.tail_stream.cuda_stream if self.tail_stream is not None else None,
                               _ptr(self.q0_mark) if self.q0_mark is not None else None, None, 0, 0)
        # second shadow buffer of W_p1t (include/ltg.h: ltg_pipe.shadow_out): the weight update of a call writes it and the two are exchanged
        # after the call, so the update starts beside the dh2 product instead of behind it.  LTGAN_SHADOW_PINGPONG=0: in place.
        self.shadow = (torch.zeros_like(engine.g_shadow) if engine.g_shadow is not None and os.environ.get("LTGAN_SHADOW_PINGPONG", "1") != "0"
                       else None)
        self.c.shadow_out = _ptr(self.shadow) if self.shadow is not None else None
        self.ahead = None           # (uitem address, n_unique, q0_ord, seq) of the call whose rows the last call brought up to date
        self.ahead_calls = 0        # calls that launched no catch-up of their own
        self.probed_for = None      # the caller's stream the side stream was last tested against (Engine._pipe_ready)
        self.handover = None        # "device-words" | "events"
        engine._pipes.add(self)     # Engine.check_pipes(): nothing reads the model out behind a wait that gave up

    def new_side_stream(self):
        self.side_stream = torch.cuda.Stream(self.sync.device)
        self.c.side_stream = self.side_stream.cuda_stream

    def new_tail_stream(self, drop=False):
        self.tail_stream = None if drop else torch.cuda.Stream(self.sync.device)
        self.c.tail_stream = self.tail_stream.cuda_stream if self.tail_stream is not None else None

    def buffers(self):
        return [self.h1pre, self.rowpart_all, self.dh2]

    def expired_waits(self):
        """device-side waits of the hand-overs that gave up (must be 0; synchronises).  Non-zero = the pipe is POISONED: every kernel of the
        one-call step that writes h2 or the model returns at once (csrc: ltg_poisoned), so the model stays what it was when the wait expired."""
        return int(self.sync[2].item())

    def reset(self):
        """after a failed call: nothing in flight, every word zero, ordinals restart (a call that stopped between its gate launches would
        otherwise leave the next one waiting for words nobody sets)"""
        torch.cuda.synchronize(self.sync.device)
        self.sync.zero_()
        if self.q0_mark is not None:
            self.q0_mark.zero_()
        self.c.seq = 0
        self.ahead = None
        self.c.next_uitem, self.c.next_nu, self.c.caught_up = None, 0, 0
        torch.cuda.synchronize(self.sync.device)


class DFork:
    """Caller-owned aux stream and device words of ltg_d_step's fork (include/ltg.h: ltg_d_opts.aux_stream / sync / seq): jobs B / C of the
    discriminator's backward run beside the critical chain of the step.  The two streams must be concurrent (tested once per caller
    stream with ltg_g_pipe_probe; otherwise the step stays on one stream)."""

    def __init__(self, engine):
        self.stream = torch.cuda.Stream(engine.device)
        self.sync = torch.zeros(16, dtype=torch.int32, device=engine.device)
        self.seq = 0
        self.probed_for = None
        self.ok = False

    def ready(self, engine, st):
        if self.probed_for != st:
            self.ok = False
            if hasattr(engine.lib, "ltg_g_pipe_probe"):
                for _ in range(4):
                    probe = cabi.ltg_pipe(self.stream.cuda_stream, None, None, None, None, None, None, 0, 0, _ptr(self.sync), None)
                    rc = engine.lib.ltg_g_pipe_probe(C.byref(probe), st)
                    if rc < 0:
                        cabi.check(rc, "ltg_g_pipe_probe")
                    if rc == 1:
                        self.ok = True
                        break
                    self.stream = torch.cuda.Stream(engine.device)
            self.sync.zero_()
            self.seq = 0
            self.probed_for = st
        return self.ok

    def expired_waits(self):
        return int(self.sync[2].item())


# arithmetic of the fp32 discriminator's GEMMs (ltg_config.d_arith): the six-term bf16 split is fp32-accurate (every fp32-path parity test passes at
# its fp32 bound with it: tests/test_gpu_parity.py::test_d_step_parity) and 3.9 ms per epoch faster on Askubuntu_Sample (profiles/r6_ab_d_arith.txt);
# "fp32" = v_mfma_f32_16x16x4_f32, the exact fma chain, stays selectable
D_ARITH_DEFAULT = "bf16x6"


class Engine:
    def __init__(self, n_items, h_sizes=(100, 150, 250, 300), lr=1e-4, p_dims=None, feature_len=None,
                 precision="bf16", seed=98765, d_seed=0, device="cuda:0", beta1=0.9, beta2=0.999, eps=1e-8,
                 item_lo=0, item_hi=None, d_precision="fp32", lazy_q0=None, q0_period=32, d_arith=None):
        """n_items = GLOBAL item count; [item_lo, item_hi) = the slab this rank owns (default: everything).
        precision: operands of the three decoder GEMMs; d_precision: operands of the discriminator GEMMs
        ("fp32" = the reference's arithmetic, "bf16", "fp8" = BASELINE config 5).
        d_arith (d_precision "fp32", config.ini-sized layers): how the fp32 products are formed -- "fp32" = exact fp32 MFMA, "bf16x6" = the
        fp32-accurate six-term bf16 split on the bf16 matrix pipe, "bf16x4" = the four-term split (2^-17 per product, opt-in); None = D_ARITH_DEFAULT
        (environment LTGAN_D_ARITH overrides; include/ltg.h: ltg_config.d_arith).
        lazy_q0: lazy Adam clock of W_q0 (include/ltg.h, ltg_gen_state.q0_last; same results as the dense sweep).  None = on
        for item slabs of 8192 items or more.  Rows are brought up to date by every forward that reads them; `g_flush()`
        does it for all rows and runs after every G step unless a trainer holds `q0_defer` for the length of its phase."""
        self.lib = cabi.load()
        import weakref
        self._pipes = weakref.WeakSet()                              # the Pipe objects created for this engine (check_pipes)
        self.check_on_flush = True                                   # False: a probe that times the host's issue rate (scripts/host_bound_probe.py)
        self.d_fork = os.environ.get("LTGAN_D_FORK", "1") != "0"     # ltg_d_step: jobs B / C of the backward on an aux stream (measurement switch)
        self._dfork = None
        self._pinned = None                                          # pin_stream()
        self.device = _require_gpu(device)
        torch.cuda.set_device(self.device)
        p_dims = p_dims or [200, 600, n_items]                       # generator.py:13
        assert p_dims[-1] == n_items
        self.I_global = n_items
        self.item_lo = int(item_lo)
        self.item_hi = int(n_items if item_hi is None else item_hi)
        self.sharded = (self.item_lo, self.item_hi) != (0, n_items)
        self.I, self.H, self.Z = self.item_hi - self.item_lo, p_dims[1], p_dims[0]   # self.I = LOCAL item count
        self.h0, self.h1, self.h2, self.h3 = h_sizes
        self.feature_len = feature_len or n_items
        self.precision = {"bf16": cabi.LTG_PREC_BF16, "fp32": cabi.LTG_PREC_FP32}[precision]
        self.d_precision = {"fp32": cabi.LTG_PREC_FP32, "bf16": cabi.LTG_PREC_BF16, "fp8": cabi.LTG_PREC_FP8}[d_precision]
        if d_arith is None:
            d_arith = os.environ.get("LTGAN_D_ARITH", D_ARITH_DEFAULT)
        self.d_arith = d_arith
        d_arith_code = {"fp32": cabi.LTG_DARITH_FP32, "bf16x6": cabi.LTG_DARITH_BF16X6, "bf16x4": cabi.LTG_DARITH_BF16X4}[d_arith]
        d_arith_code |= (int(os.environ.get("LTGAN_D_ARITH_SET", "0"), 0) & 15) << 4     # measurement switch: which of the four GEMM kernels
        self.cfg = cabi.ltg_config(self.I, self.H, self.Z, self.feature_len, self.h0, self.h1, self.h2, self.h3,
                                   self.precision, 0, self.item_lo, n_items if self.sharded else 0, self.d_precision, d_arith_code,
                                   lr, beta1, beta2, eps, seed)
        self.cfg.tuning = int(os.environ.get("LTGAN_TUNING", "0"), 0)     # measurement switch: ltg_config.tuning (include/ltg.h)
        self.lr, self.beta1, self.beta2 = lr, beta1, beta2
        self.adam_t = 0                                              # shared by D and G (Q5)
        if lazy_q0 is None:
            lazy_q0 = os.environ.get("LTGAN_LAZY_Q0", "1") != "0"   # measurement switch: 0 = dense sweep every step
        self.lazy_q0 = bool(lazy_q0) and self.I >= 8192
        self.q0_period = int(q0_period)
        if self.lazy_q0 and not 1 <= self.q0_period <= cabi.LTG_Q0_HIST // 2:
            raise ValueError("q0_period must be in [1, %d] (the clock's ring keeps the last %d step sizes)" % (cabi.LTG_Q0_HIST // 2, cabi.LTG_Q0_HIST))
        self.q0_defer = False                                        # True: the caller flushes (end of its G phase)
        self.q0_sweep_overlap = os.environ.get("LTGAN_Q0_OVERLAP", "1") != "0"   # measurement switch
        self._q0_dirty = False
        self._init_generator(seed)
        self._init_discriminator(d_seed)
        self._ws = None
        self._ws_key = (0, 0)
        self.loss_buf = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.overlap = True                                          # fake-tower forward on a second stream (ltg_g_opts.aux_stream)
        self._aux = None

    # ------------------------------------------------------------------ parameters
    def _init_generator(self, seed):
        self.set_generator(init_generator_host(self.I_global, self.H, self.Z, seed))     # global tables: set_generator keeps this rank's slab

    def set_generator(self, arrays, m=None, v=None):
        """arrays in engine layout: [W_q0 [I,H], W_q1 [H,2Z], W_p0 [Z,H], W_p1t [I,H], b_q0, b_q1, b_p0, b_p1] with
        I = the GLOBAL item count; an item-sharded engine keeps rows [item_lo, item_hi) of the item tables."""
        dev = self.device
        lo, hi = self.item_lo, self.item_hi

        def cut(seq):
            seq = list(seq)
            if seq[0].shape[0] == self.I_global and self.sharded:
                for i in (0, 3, 7):
                    seq[i] = seq[i][lo:hi]
            return seq
        arrays = cut(arrays)
        m = cut(m) if m is not None else None
        v = cut(v) if v is not None else None
        self.g_p = [torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev).contiguous() for a in arrays]
        self.g_m = [torch.zeros_like(t) if m is None else torch.as_tensor(np.ascontiguousarray(m[i]), dtype=torch.float32).to(dev)
                    for i, t in enumerate(self.g_p)]
        self.g_v = [torch.zeros_like(t) if v is None else torch.as_tensor(np.ascontiguousarray(v[i]), dtype=torch.float32).to(dev)
                    for i, t in enumerate(self.g_p)]
        if self.I >= 8192 and os.environ.get("LTGAN_ONE_BLOCK", "1") != "0":
            # theta / m / v of W_p1t in ONE allocation (each table on a 2-MiB boundary): the weight update streams the three in lock step, and
            # a sweep over three separate allocations measured 2-6 % below the same sweep over one block (scripts/micro/stagger.hip: 5.56-5.79
            # against 5.90 TB/s at 200 000 items; which of the two a process got showed as the two modes of profiles/r4_c4_two_modes.txt)
            n = self.g_p[3].numel()
            per = ((n * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // 4
            self._p1_block = torch.empty(3 * per, dtype=torch.float32, device=dev)
            for k, lst in enumerate((self.g_p, self.g_m, self.g_v)):
                view = self._p1_block[k * per:k * per + n].view_as(lst[3])
                view.copy_(lst[3])
                lst[3] = view
        arr = lambda ts: (cabi.vp * 8)(*[_ptr(t) for t in ts])
        # bf16 shadow of W_p1t for the streaming decoder kernels (large item slabs only)
        self.g_shadow = None
        if self.precision == cabi.LTG_PREC_BF16 and self.I >= 8192 and self.I % 8 == 0 and self.H <= 608:
            self.g_shadow = torch.zeros(self.I, 608, dtype=torch.int16, device=dev)
        self.q0_last = self.q0_lr_hist = None
        if self.lazy_q0:                                             # all rows current at ordinal 0
            self.q0_last = torch.zeros(self.I, dtype=torch.int32, device=dev)
            self.q0_lr_hist = torch.zeros(cabi.LTG_Q0_HIST, dtype=torch.float32, device=dev)
        self._q0_dirty = False
        self._forget_ahead()                                         # (new weights, ordinal 0: no catch-up announced against the old clock stays valid)
        self.gen_c = cabi.ltg_gen_state(arr(self.g_p), arr(self.g_m), arr(self.g_v), _ptr(self.g_shadow),
                                        _ptr(self.q0_last), _ptr(self.q0_lr_hist), 0, self.q0_period if self.lazy_q0 else 0)
        if self.g_shadow is not None:
            cabi.check(self.lib.ltg_refresh_shadow(C.byref(self.cfg), C.byref(self.gen_c), self.stream()), "ltg_refresh_shadow")

    def _init_discriminator(self, seed):
        emb, host = init_discriminator_host(self.feature_len, (self.h0, self.h1, self.h2, self.h3), seed)
        self.set_discriminator(emb, host)

    def resize_discriminator(self, h_sizes, feature_len=None, d_seed=0):
        """Re-create the discriminator part for other layer sizes (discriminator.py:3: the sizes are arguments of the
        factory, not of the generator): fresh truncated-normal variables, zero Adam moments, new workspace."""
        self.h0, self.h1, self.h2, self.h3 = (int(x) for x in h_sizes)
        if feature_len is not None:
            self.feature_len = int(feature_len)
        self.cfg.d_feat, self.cfg.d_h0, self.cfg.d_h1, self.cfg.d_h2, self.cfg.d_h3 = self.feature_len, self.h0, self.h1, self.h2, self.h3
        self._init_discriminator(d_seed)
        self._ws, self._ws_key = None, (0, 0)

    def set_learning_rate(self, lr):
        """tf.train.AdamOptimizer(LEARNING_RATE) (train.py:160): one optimiser, one rate for the D and the G update."""
        self.lr = float(lr)
        self.cfg.lr = self.lr

    def set_discriminator(self, emb, arrays, m=None, v=None):
        dev = self.device
        self.d_emb = torch.as_tensor(np.ascontiguousarray(emb), dtype=torch.float32).to(dev).contiguous()
        host = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        host[6] = host[6].reshape(-1)                                # w4 [h3,1] -> [h3]
        host[7] = host[7].reshape(-1)
        # the eight trainable tensors (and each of their Adam moments) live back to back in ONE buffer, padded to whole float4:
        # the library then runs the optimiser as a single flat 16-byte-per-lane sweep; d_p / d_m / d_v are views into it
        sizes = [a.size for a in host]
        total = (sum(sizes) + 3) // 4 * 4

        def flat(arrs):
            buf = torch.zeros(total, dtype=torch.float32, device=dev)
            views, off = [], 0
            for a, ref in zip(arrs, host):
                n = ref.size
                if a is not None:
                    buf[off:off + n] = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).reshape(-1).to(dev)
                views.append(buf[off:off + n].view(ref.shape))
                off += n
            return buf, views
        self.d_flat_p, self.d_p = flat(host)
        self.d_flat_m, self.d_m = flat([None] * 8 if m is None else list(m))
        self.d_flat_v, self.d_v = flat([None] * 8 if v is None else list(v))
        arr = lambda ts: (cabi.vp * 8)(*[_ptr(t) for t in ts])
        # e4m3 operand-format shadows for the fp8 mode of the wide discriminator (every layer size a multiple of 64): the frozen
        # embedding table and the TRANSPOSED weights, kept in step by the library's Adam sweep
        self.d_fp8 = None
        h0, h1, h2, h3 = self.h0, self.h1, self.h2, self.h3
        if self.d_precision == cabi.LTG_PREC_FP8 and all(x % 64 == 0 for x in (h0, h1, h2, h3)):
            u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
            self.d_fp8 = (u8(self.feature_len * h0), u8(h1 * h0), u8(h2 * h0), u8(h3 * (h1 + h2)), u8((h1 + h2) * h3))
        sh = [_ptr(t) for t in self.d_fp8] if self.d_fp8 else [None] * 5
        self.disc_c = cabi.ltg_disc_state(_ptr(self.d_emb), arr(self.d_p), arr(self.d_m), arr(self.d_v), *sh)
        if self.d_fp8:
            cabi.check(self.lib.ltg_refresh_d_shadow(C.byref(self.cfg), C.byref(self.disc_c), self.stream()), "ltg_refresh_d_shadow")

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        """raw handle of the stream the library launches on: torch's current stream -- or the one a trainer pinned for the length
        of a phase (`pin_stream`: the lookup costs a few microseconds and a sharded G step makes seven of them)"""
        return self._pinned if self._pinned is not None else torch.cuda.current_stream(self.device).cuda_stream

    def pin_stream(self, on=True):
        self._pinned = torch.cuda.current_stream(self.device).cuda_stream if on else None

    def workspace(self, rows, pairs):
        rows, pairs = max(1, int(rows)), max(1, int(pairs))
        if self._ws is None or rows > self._ws_key[0] or pairs > self._ws_key[1]:
            rows = max(rows, self._ws_key[0])
            pairs = max(pairs, self._ws_key[1])
            nbytes = self.lib.ltg_workspace_bytes(C.byref(self.cfg), rows, pairs)
            if nbytes == 0:
                raise cabi.LtgError("ltg_workspace_bytes rejected the configuration")
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_key = (rows, pairs)
        return self._ws

    def _fwd_scratch(self, rows):
        """row-statistics scratch of the forward (large item slabs only)"""
        if self.I <= 8192:
            return None
        need = int(self.lib.ltg_forward_scratch_bytes(C.byref(self.cfg), rows))
        if getattr(self, "_fs", None) is None or self._fs.numel() < need:
            self._fs = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._fs

    def new_acts(self, rows):
        return Acts(rows, self.I, self.H, self.Z, self.device)

    def _fork_handles(self):
        if not self.overlap:
            return None, None, None
        if self._aux is None:
            from ._hip import EventPair
            self._aux = (torch.cuda.Stream(self.device), EventPair(timing=False), EventPair(timing=False))
        st, ev = self._aux[:2]
        return st.cuda_stream, ev.start, ev.stop

    def _sweep_event(self):
        """ltg_g_opts.ev_sweep: the lazy clock's rotating slice runs on the aux stream beside the decoder kernels"""
        if not (self.overlap and self.lazy_q0 and self.q0_sweep_overlap):
            return None
        self._fork_handles()
        return self._aux[2].start

    def next_adam_t(self):
        self.adam_t += 1
        return self.adam_t

    # ------------------------------------------------------------------ the five run signatures
    def forward(self, batch, acts, keep_prob=0.75, is_training=0.0, rng_step=0, probs_out=None, drop_keep=None, eps=None,
                probe=None, rows_per_step=0):
        """sess.run(generator_out, {input_ph: X})  -- train.py:200, :339; test.py:146.
        rows_per_step > 0: `batch` spans several batches of that many rows; batch k draws its dropout with rng_step + k."""
        assert acts.rows >= batch.n_rows
        o = cabi.ltg_fwd_opts(keep_prob, is_training, rng_step, _ptr(drop_keep), _ptr(eps), _pp(probe), int(rows_per_step), 0)
        ws = self._fwd_scratch(batch.n_rows)
        rc = self.lib.ltg_vae_forward(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(o),
                                      C.byref(acts.c), _ptr(probs_out), _ptr(ws), ws.numel() if ws is not None else 0, self.stream())
        cabi.check(rc, "ltg_vae_forward")

    def sample_pairs(self, samp_c, acts, gen_out, pop_out, cnt_out, out_off=0):
        """the Python loop train.py:212-251 (+ sample.py:40-67) on the device."""
        rc = self.lib.ltg_sample_pairs(C.byref(self.cfg), C.byref(samp_c), _ptr(acts.logits), _ptr(acts.lse),
                                       _ptr(gen_out, out_off), _ptr(pop_out, out_off), _ptr(cnt_out), self.stream())
        cabi.check(rc, "ltg_sample_pairs")

    def d_step(self, real, fake, keep_prob=0.7, rng_step=0, loss_out=None, drop_real=None, drop_fake=None, probe=None):
        """sess.run([d_trainer, d_loss_mean], ...)  -- train.py:300."""
        loss_out = self.loss_buf if loss_out is None else loss_out
        ws = self.workspace(1, real.n + fake.n)
        dr = (cabi.vp * 3)(*[_ptr(t) for t in (drop_real or (None, None, None))])
        df = (cabi.vp * 3)(*[_ptr(t) for t in (drop_fake or (None, None, None))])
        o = cabi.ltg_d_opts(keep_prob, self.next_adam_t(), rng_step, dr, df, _pp(probe))
        if self.d_fork and self.d_precision == cabi.LTG_PREC_FP32:
            if self._dfork is None:
                self._dfork = DFork(self)
            if self._dfork.ready(self, self.stream()):
                self._dfork.seq = (self._dfork.seq + 1) & 0xFFFFFFFF
                o.aux_stream, o.sync, o.seq = self._dfork.stream.cuda_stream, _ptr(self._dfork.sync), self._dfork.seq
        rc = self.lib.ltg_d_step(C.byref(self.cfg), C.byref(self.disc_c), C.byref(real.c), C.byref(fake.c), C.byref(o),
                                 _ptr(loss_out), _ptr(ws), ws.numel(), self.stream())
        cabi.check(rc, "ltg_d_step")
        return loss_out

    # ------------------------------------------------------------------ the D step cut at its exchange point (pair rows split)
    def d_grad_floats(self):
        return int(self.lib.ltg_d_grad_floats(C.byref(self.cfg)))

    def d_grad(self, real, fake, row_lo, row_hi, grad_out, keep_prob=0.7, rng_step=0, drop_real=None, drop_fake=None):
        """forward + backward over pair rows [row_lo, row_hi) of real | fake -> this rank's gradient vector (+ loss share)"""
        ws = self.workspace(1, max(1, row_hi - row_lo))
        dr = (cabi.vp * 3)(*[_ptr(t) for t in (drop_real or (None, None, None))])
        df = (cabi.vp * 3)(*[_ptr(t) for t in (drop_fake or (None, None, None))])
        o = cabi.ltg_d_opts(keep_prob, 0, rng_step, dr, df, None)
        rc = self.lib.ltg_d_grad(C.byref(self.cfg), C.byref(self.disc_c), C.byref(real.c), C.byref(fake.c), int(row_lo), int(row_hi),
                                 C.byref(o), _ptr(grad_out), _ptr(ws), ws.numel(), self.stream())
        cabi.check(rc, "ltg_d_grad")

    def d_apply(self, grad, loss_out=None):
        """the (all-reduced) gradient vector -> one TF-Adam sweep at the next shared step; d_loss -> loss_out[0]"""
        loss_out = self.loss_buf if loss_out is None else loss_out
        rc = self.lib.ltg_d_apply(C.byref(self.cfg), C.byref(self.disc_c), _ptr(grad), self.next_adam_t(), _ptr(loss_out), self.stream())
        cabi.check(rc, "ltg_d_apply")
        return loss_out

    def fake_tower_batched(self, fake, seg_of, seg_row0, seg_step, y_out, d_keep_prob=0.7, seg_off=0, y_off=0):
        """y_generated of MANY pair batches in one pass (the discriminator is fixed during phase G): `fake` = the concatenated
        slots, seg_of (+ seg_off) / seg_row0 / seg_step as in include/ltg.h; the G steps then take their slice through `y_pre`."""
        ws = self.workspace(1, fake.n)
        cabi.check(self.lib.ltg_fake_tower_batched(C.byref(self.cfg), C.byref(self.disc_c), C.byref(fake.c), _ptr(seg_of, seg_off), _ptr(seg_row0),
                                                   _ptr(seg_step), d_keep_prob, _ptr(y_out, y_off), _ptr(ws), ws.numel(), self.stream()),
                   "ltg_fake_tower_batched")

    def g_step(self, batch, fake, acts, cnt, anneal, gan_lambda=1.0, keep_prob=0.75, is_training=1.0, d_keep_prob=0.7,
               rng_step=0, d_rng_step=0, loss_out=None, drop_keep=None, eps=None, drop_fake=None, probe=None, y_pre=None, y_off=0):
        """sess.run([g_trainer, g_loss_mean, g_vae_loss, gan_loss], ...)  -- train.py:326.
        y_pre (+ y_off): y_generated of this batch's fake pairs from fake_tower_batched (same d_rng_step): no tower in the step."""
        loss_out = self.loss_buf if loss_out is None else loss_out
        ws = self.workspace(batch.n_rows, fake.n)
        f = cabi.ltg_fwd_opts(keep_prob, is_training, rng_step, _ptr(drop_keep), _ptr(eps), _pp(probe))
        df = (cabi.vp * 3)(*[_ptr(t) for t in (drop_fake or (None, None, None))])
        o = cabi.ltg_g_opts(f, anneal, gan_lambda, d_keep_prob, self.next_adam_t(), d_rng_step, df, _ptr(cnt), _pp(probe),
                            *self._fork_handles(), 0, 0, self._sweep_event(), _ptr(y_pre, y_off))
        rc = self.lib.ltg_g_step(C.byref(self.cfg), C.byref(self.gen_c), C.byref(self.disc_c), C.byref(batch.c),
                                 C.byref(fake.c), C.byref(o), C.byref(acts.c), _ptr(loss_out), _ptr(ws), ws.numel(),
                                 self.stream())
        cabi.check(rc, "ltg_g_step")
        self._q0_stepped()
        return loss_out

    def _q0_stepped(self):
        """one more G step on the lazy clock of W_q0 (ltg_gen_state.q0_ord is the caller's to advance)"""
        if self.lazy_q0:
            self.gen_c.q0_ord += 1
            self._q0_dirty = True
            if not self.q0_defer:
                self.g_flush()

    def g_flush(self):
        """every deferred zero-gradient Adam step of W_q0, all rows: before W_q0 / its moments are read outside a forward"""
        if self.lazy_q0 and self._q0_dirty:
            if self.check_on_flush:
                self.check_pipes()                                   # (end of a G phase / before the model is read out: one host sync.  BEFORE the
                                                                     # flush: behind a wait that gave up the clock's ordinals are not to be trusted)
            cabi.check(self.lib.ltg_g_flush(C.byref(self.cfg), C.byref(self.gen_c), self.stream()), "ltg_g_flush")
            self._q0_dirty = False
            # every row is current: restart the ordinals (the int32 clock never grows without bound; stream order keeps the
            # memset behind the flush kernel)
            self.q0_last.zero_()
            self.gen_c.q0_ord = 0
            self._forget_ahead()       # (an announcement made against the old ordinals must not match a later call's key)

    def _forget_ahead(self):
        """the clock's ordinals restart (flush, new weights): no pipe's catch-up-ahead announcement is valid any more"""
        for pipe in list(self._pipes):
            pipe.ahead = None
            pipe.c.next_uitem, pipe.c.next_nu, pipe.c.caught_up = None, 0, 0

    def check_pipes(self):
        """raises if a device-side wait of a one-call G step gave up (ltg_pipe.sync[2]).  The kernels behind such a wait have skipped
        their work, so the model is the one from before that step -- but the run is not the run that was asked for.  ONE device-to-host
        copy and one synchronisation however many pipes the engine has."""
        words = [pipe.sync[2:3] for pipe in list(self._pipes) + ([self._dfork] if self._dfork is not None else [])]
        if not words:
            return
        n = int((words[0] if len(words) == 1 else torch.cat(words)).sum().item())
        if n:
            torch.cuda.synchronize(self.device)
            self.refresh_shadow()      # (a skipped weight update did not write the shadow buffer the host has switched to since)
            raise cabi.LtgError("%d device-side wait(s) of a step's hand-overs gave up: the steps behind them were skipped, "
                                "the results of that phase are not trustworthy" % n)

    # ------------------------------------------------------------------ the G step cut at its exchange points
    def fwd_opts(self, keep_prob=0.75, is_training=0.0, rng_step=0, drop_keep=None, eps=None, probe=None):
        return cabi.ltg_fwd_opts(keep_prob, is_training, rng_step, _ptr(drop_keep), _ptr(eps), _pp(probe))

    def g_opts(self, cnt, anneal, gan_lambda=1.0, keep_prob=0.75, is_training=1.0, d_keep_prob=0.7, rng_step=0, d_rng_step=0,
               drop_keep=None, eps=None, drop_fake=None, probe=None, y_pre=None, y_off=0):
        f = self.fwd_opts(keep_prob, is_training, rng_step, drop_keep, eps, probe)
        df = (cabi.vp * 3)(*[_ptr(t) for t in (drop_fake or (None, None, None))])
        return cabi.ltg_g_opts(f, anneal, gan_lambda, d_keep_prob, self.next_adam_t(), d_rng_step, df, _ptr(cnt), _pp(probe),
                               *self._fork_handles(), 0, 0, None, _ptr(y_pre, y_off))      # ltg_g_bwd_rest forks / joins inside the call

    # ------------------------------------------------------------------ the item-sharded G step as ONE call
    def sharded_step_ok(self, rows):
        """ltg_g_step_sharded serves this engine's configuration (bf16 + shadow, lazy clock, slab of >= 8192 items, <= 128 rows)"""
        if not hasattr(self.lib, "ltg_g_step_sharded_ok"):      # an older build loaded through LTG_HIP_LIB (A/B timing)
            return False
        return bool(self.lib.ltg_g_step_sharded_ok(C.byref(self.cfg), C.byref(self.gen_c), int(rows)))

    def g_step_sharded(self, batch, fake, acts, gopts, pipe, comm=None, loss_out=None, next_batch=None):
        """every launch of the step and its exchanges from one call (comm: _rccl.RcclComm / HostComm; None = one rank).
        next_batch: the batch of the NEXT call on this pipe when the caller knows it -- its rows of W_q0 are brought up to the lazy clock
        during this call on the side stream, and the next call launches no catch-up (include/ltg.h: catch-up ahead)"""
        loss_out = self.loss_buf if loss_out is None else loss_out
        ws = self.workspace(batch.n_rows, fake.n)
        self._pipe_ready(pipe)
        if pipe.c.seq >= 0x7FFFFFF0:                                   # (the marks compare ordinals: restart long before they could repeat)
            self.pipe_join(pipe)
            pipe.reset()
        pipe.c.seq = (pipe.c.seq + 1) & 0xFFFFFFFF                     # the call's ordinal on this pipe (what its gates count in)
        key = (batch.c.uitem, int(batch.c.n_unique), int(self.gen_c.q0_ord), int(pipe.c.seq))
        pipe.c.caught_up = 1 if (pipe.ahead is not None and pipe.ahead == key) else 0
        pipe.ahead_calls += pipe.c.caught_up
        pipe.ahead = None
        pipe.c.next_uitem, pipe.c.next_nu = None, 0
        plan = (self.lib.ltg_g_step_sharded_plan(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(pipe.c))
                if hasattr(self.lib, "ltg_g_step_sharded_plan") else 0)      # (an ABI 11 / 12 build under LTG_AB_COMPAT: no catch-up ahead, one shadow buffer)
        if next_batch is not None and next_batch.c.uitem and self.q0_defer and (plan & cabi.LTG_PLAN_AHEAD):
            pipe.c.next_uitem, pipe.c.next_nu = next_batch.c.uitem, int(next_batch.c.n_unique)
            pipe.ahead = (next_batch.c.uitem, int(next_batch.c.n_unique), int(self.gen_c.q0_ord) + 1, (int(pipe.c.seq) + 1) & 0xFFFFFFFF)
        rc = self.lib.ltg_g_step_sharded(C.byref(self.cfg), C.byref(self.gen_c), C.byref(self.disc_c), C.byref(batch.c), C.byref(fake.c),
                                         C.byref(gopts), C.byref(acts.c), C.byref(comm.c) if comm is not None else None, C.byref(pipe.c),
                                         _ptr(loss_out), _ptr(ws), ws.numel(), self.stream())
        if rc != 0:
            pipe.reset()               # (the call may have stopped between its gate launches: words of this ordinal that nobody will set)
            self.refresh_shadow()      # (... or between the update's launch and the exchange below)
        cabi.check(rc, "ltg_g_step_sharded")
        if plan & cabi.LTG_PLAN_SHADOW:    # the update wrote the pipe's buffer: it is the shadow from now on, the old one the next call's target
            self.g_shadow, pipe.shadow = pipe.shadow, self.g_shadow
            self.gen_c.wp1t_bf16, pipe.c.shadow_out = _ptr(self.g_shadow), _ptr(pipe.shadow)
        if not self.q0_defer:
            self.pipe_join(pipe)       # a standalone step: nothing stays in flight behind the call (a trainer joins once per phase)
        self._q0_stepped()             # (standalone: flushes the clock and checks the pipe's poison -- g_flush)
        return loss_out

    def refresh_shadow(self):
        """the bf16 shadow of W_p1t rebuilt from the fp32 rows (after loading weights; after a failed pipelined call)"""
        if self.g_shadow is not None:
            cabi.check(self.lib.ltg_refresh_shadow(C.byref(self.cfg), C.byref(self.gen_c), self.stream()), "ltg_refresh_shadow")

    def _pipe_ready(self, pipe):
        """The device-word hand-over of ltg_g_step_sharded needs two CONCURRENT streams; HIP maps streams onto a few hardware queues, so
        the pair is tested once (ltg_g_pipe_probe; a few other side streams are tried, then the pipe falls back to event pairs)."""
        st = self.stream()
        if pipe.probed_for == st:
            return
        if pipe.probed_for is not None:
            self.pipe_join(pipe)                     # the mode may change: nothing in flight across the change
            torch.cuda.synchronize(self.device)
        if pipe.c.flags & cabi.LTG_PIPE_NO_DEC1_FORK:
            pipe.handover = "none (everything in program order)"
        elif pipe.c.flags & cabi.LTG_PIPE_EVENTS:
            pipe.handover = "events"
        else:
            pipe.handover = "events"

            def probe():
                ok = self.lib.ltg_g_pipe_probe(C.byref(pipe.c), st) if hasattr(self.lib, "ltg_g_pipe_probe") else 0   # (an older build under LTG_AB_COMPAT)
                if ok < 0:
                    cabi.check(ok, "ltg_g_pipe_probe")
                return ok == 1
            tail = pipe.c.tail_stream
            pipe.c.tail_stream = None                # the side stream first ...
            for _ in range(6):
                if probe():
                    pipe.handover = "device-words"
                    break
                pipe.new_side_stream()
            if pipe.handover == "events":
                pipe.c.flags |= cabi.LTG_PIPE_EVENTS
                pipe.new_tail_stream(drop=True)
            elif tail:                               # ... then the tail's own stream: a third concurrent queue, or the tail stays on the caller's stream
                pipe.c.tail_stream = tail
                for k in range(10):     # (a process that has created many streams: the next few may share the side stream's queue)
                    if probe():
                        break
                    pipe.new_tail_stream(drop=(k == 9))
                if pipe.tail_stream is not None:
                    pipe.handover = "device-words + tail stream"
        pipe.probed_for = st

    def pipe_join(self, pipe):
        cabi.check(self.lib.ltg_g_pipe_join(C.byref(pipe.c), self.stream()), "ltg_g_pipe_join")

    def g_fwd_enc(self, batch, acts, fopts):
        cabi.check(self.lib.ltg_g_fwd_enc(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(fopts), C.byref(acts.c),
                                          self.stream()), "ltg_g_fwd_enc")

    def g_fwd_rest(self, batch, fake, acts, fopts, rowpart_out):
        ws = self.workspace(batch.n_rows, fake.n if fake else 1)
        cabi.check(self.lib.ltg_g_fwd_rest(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(fake.c) if fake else None,
                                           C.byref(fopts), C.byref(acts.c), _ptr(rowpart_out), _ptr(ws), ws.numel(), self.stream()),
                   "ltg_g_fwd_rest")

    def rowstats_combine(self, rowpart_all, n_ranks, n_rows, lse_out):
        ws = self.workspace(n_rows, 1)
        cabi.check(self.lib.ltg_rowstats_combine(C.byref(self.cfg), _ptr(rowpart_all), n_ranks, n_rows, _ptr(lse_out), _ptr(ws), ws.numel(),
                                                 self.stream()), "ltg_rowstats_combine")

    def g_bwd_dec(self, batch, fake, acts, gopts, rowpart_all, n_ranks, loss_out, dh2_out):
        ws = self.workspace(batch.n_rows, fake.n)
        cabi.check(self.lib.ltg_g_bwd_dec(C.byref(self.cfg), C.byref(self.gen_c), C.byref(self.disc_c), C.byref(batch.c), C.byref(fake.c),
                                          C.byref(gopts), C.byref(acts.c), _ptr(rowpart_all), n_ranks, _ptr(loss_out), _ptr(dh2_out),
                                          _ptr(ws), ws.numel(), self.stream()), "ltg_g_bwd_dec")

    def g_fake_tower(self, batch, fake, gopts, stream=None):
        """the fake tower's forward of the step, on `stream` (a torch stream; default: the current one); sets gopts.fake_done"""
        ws = self.workspace(batch.n_rows, fake.n)
        st = self.stream() if stream is None else stream.cuda_stream
        cabi.check(self.lib.ltg_g_fake_tower(C.byref(self.cfg), C.byref(self.disc_c), C.byref(fake.c), C.byref(gopts), batch.n_rows, _ptr(ws),
                                             ws.numel(), st), "ltg_g_fake_tower")
        gopts.fake_done = 1

    def g_bwd_dec1(self, batch, fake, acts, gopts, stream=None):
        """Adam on the local W_p1t / b_p1 rows: needs nothing of the dh2 exchange, so it is issued while that all-reduce flies
        (`stream`: a torch side stream -- the caller orders it behind ltg_g_bwd_dec and joins it before the next forward)"""
        ws = self.workspace(batch.n_rows, fake.n)
        st = self.stream() if stream is None else stream.cuda_stream
        cabi.check(self.lib.ltg_g_bwd_dec1(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(fake.c), C.byref(gopts),
                                           C.byref(acts.c), _ptr(ws), ws.numel(), st), "ltg_g_bwd_dec1")
        gopts.dec1_done = 1

    def g_bwd_rest(self, batch, fake, acts, gopts, dh2):
        ws = self.workspace(batch.n_rows, fake.n)
        cabi.check(self.lib.ltg_g_bwd_rest(C.byref(self.cfg), C.byref(self.gen_c), C.byref(batch.c), C.byref(fake.c), C.byref(gopts),
                                           C.byref(acts.c), _ptr(dh2), _ptr(ws), ws.numel(), self.stream()), "ltg_g_bwd_rest")
        self._q0_stepped()

    def gather_cand_logits(self, samp_c, acts, out):
        cabi.check(self.lib.ltg_gather_cand_logits(C.byref(self.cfg), C.byref(samp_c), _ptr(acts.logits), _ptr(out), self.stream()),
                   "ltg_gather_cand_logits")

    def rank_metrics(self, acts, tr, te, out, k_ndcg=100, k_r1=20, k_r2=50):
        """pred[X.nonzero()] = -inf + NDCG@100 / Recall@20 / Recall@50  -- train.py:341-346."""
        rc = self.lib.ltg_rank_metrics(C.byref(self.cfg), _ptr(acts.logits), C.byref(tr.c), C.byref(te.c), k_ndcg, k_r1,
                                       k_r2, _ptr(out), self.stream())
        cabi.check(rc, "ltg_rank_metrics")

    # ------------------------------------------------------------------ ranking metrics cut at their exchange points
    def rank_scores(self, acts, tr, te, score_out):
        cabi.check(self.lib.ltg_rank_scores(C.byref(self.cfg), _ptr(acts.logits), C.byref(tr.c), C.byref(te.c), _ptr(score_out),
                                            self.stream()), "ltg_rank_scores")

    def rank_counts(self, acts, tr, te, score, count_out):
        cabi.check(self.lib.ltg_rank_counts(C.byref(self.cfg), _ptr(acts.logits), C.byref(tr.c), C.byref(te.c), _ptr(score),
                                            _ptr(count_out), self.stream()), "ltg_rank_counts")

    def rank_finish(self, te, counts, out, k_ndcg=100, k_r1=20, k_r2=50):
        cabi.check(self.lib.ltg_rank_finish(C.byref(te.c), _ptr(counts), k_ndcg, k_r1, k_r2, _ptr(out), self.stream()), "ltg_rank_finish")

    # ------------------------------------------------------------------ views in the reference's shapes
    def generator_params_tf(self):
        """The 8 tensors in the reference's order and TF shapes (MultiVAE.py:129-141); W_p1 is a
        transposed view of the item-major storage."""
        self.g_flush()
        p = self.g_p
        return [p[0], p[1], p[2], p[3].t(), p[4], p[5], p[6], p[7]]

    def discriminator_params_tf(self):
        p = self.d_p
        return [p[0], p[1], p[2], p[3], p[4], p[5], p[6].reshape(-1, 1), p[7]]
