#!/usr/bin/env python3
"""bench.py -- train users/sec at BATCH_SIZE=100 of the Long-Tail-GAN adversarial training path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload askubuntu|ml20m|c4]

N > 1 without a launcher: bench.py starts the N rank processes itself (torch.distributed.run as a child of a parent that never
touches the GPU) and relays rank 0's JSON line; under `python -m torch.distributed.run --nproc-per-node N` it is a rank.

A "step" is one pass of the hot path over the workload's batch set: phase C (generator forward +
fake-pair sampling per batch), NUM_SUB_EPOCHS discriminator passes, NUM_SUB_EPOCHS generator passes
(train.py:192-329, config.ini defaults: BATCH_SIZE=100, NUM_EPOCH=80 -> 10 sub-epochs).  For the
default workload (Askubuntu_Sample: 10 001 users x 1 000 items, 101 batches) one step is exactly
one global epoch of the reference.  value = users processed / wall seconds, inputs resident in HBM.

Prints ONE JSON line (see README / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, help="askubuntu | ml20m | c4 | custom:<items>; default: askubuntu at 1 GPU, c4 (item-sharded) at N > 1")
    ap.add_argument("--parallelism", default=None, choices=["item-shard", "replicas"], help="default: item-shard when N > 1")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--sub-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--users", type=int, default=None, help="synthetic workloads: number of user rows to generate (default: a bounded sample)")
    ap.add_argument("--d-sizes", default="100,150,250,300",
                    help="discriminator h0,h1,h2,h3 (config.ini defaults; BASELINE config 5 = 2048,1024,512,256)")
    ap.add_argument("--d-precision", default="fp32", choices=["fp32", "bf16", "fp8"],
                    help="operand precision of the discriminator GEMMs (fp32 = the reference's arithmetic)")
    ap.add_argument("--d-arith", default=None, choices=["fp32", "bf16x6", "bf16x4"],
                    help="how the fp32 discriminator's products are formed (ltg_config.d_arith): exact fp32 MFMA, the fp32-accurate six-term bf16 split, "
                         "or the four-term split (opt-in, 2^-17 per product); default: the engine's (D_ARITH_DEFAULT)")
    ap.add_argument("--d-split", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: split the discriminator's pair rows over the ranks + gradient all-reduce (auto: when the discriminator has >= 1 M parameters)")
    ap.add_argument("--variant", type=int, default=0, help="kernel tuning knob (ltg_config.tuning)")
    ap.add_argument("--warm-moments", action="store_true", help="(the default for the synthetic workloads) give every row of W_q0 non-zero Adam moments "
                    "before timing: the state of the 136 000- / 1 000 000-user configurations, where every item has been seen -- the lazy clock "
                    "then has its full deferred arithmetic to do")
    ap.add_argument("--cold-moments", action="store_true", help="leave the moments of W_q0 at zero (a bounded sample touches a fraction of the items: "
                    "rows no batch has touched then cost the lazy clock nothing -- NOT the state of the named configurations)")
    ap.add_argument("--no-probe", action="store_true", help="skip the HIP-event kernel probes (use under rocprofv3 --pmc)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the bounded C3 / C4-shaped 1-GPU measurements reported beside the headline")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def load_workload(name, batch_size, device, users=None):
    from ltgan.dataset import DeviceData, IndexData, materialize_askubuntu
    if name == "askubuntu":
        raw = os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz")
        d = tempfile.mkdtemp(prefix="askubuntu_")
        materialize_askubuntu(raw, d)
        idx = IndexData.from_dir(d)
        desc = "Askubuntu_Sample (10001 users x 1000 items, 179368 interactions; dataset files rebuilt from tests/golden/askubuntu_raw.npz)"
    else:
        from ltgan.synthetic import synthetic_index
        idx, desc = synthetic_index(name, users=users or 6400)   # bounded sample: 64 batches of the named shape
    return idx, DeviceData(idx, batch_size, device), desc


PEAK = {"hbm": 8.0e12, "mfma_bf16": 2.5e15, "mfma_fp32": 157.3e12}   # MI355X_MICROARCH.md chip-level parameters


# probe name -> kernel names in a rocprofv3 --kernel-trace --stats summary (template arguments and namespaces stripped)
ROCPROF_KERNELS = {
    "enc0_fwd": ("fk_enc0_fwd", "k_enc0_fwd"), "enc1": ("fk_enc1",), "dec0": ("fk_dec0",), "dec1_fwd": ("fk_dec1", "k_dec1_fwd_stream", "k_dec1_fwd_stream2"),
    "row_dlogits": ("fk_row_dlogits",), "dh2": ("fk_dh2", "k_dh2_stream"), "dz": ("fk_dz", "fk_dz_dh2"), "dh1": ("fk_dh1",),
    "dec1_bwd_adam": ("k_dec1_bwd_adam_stream",), "enc0_grad": ("fk_enc0_grad_rows", "fk_enc0_grad"), "enc0_bwd_adam": ("k_q0_sweep", "k_q0_touch_ahead", "k_q0_touch_unique"),
    "g_tail": ("fk_g_tail",), "d_l1": ("fk_d_l1", "fk8t_d_l1"), "d_l2": ("fk_d_l2", "fk8s_d_l2"), "d_bwd1": ("fk_d_bwd1", "k8_d_bwd1"),
    "d_bwd2": ("fk_d_bwd2", "k8_d_bwd2"), "d_adam": ("fk_d_adam", "k8_d_adam"),
}
# the one-wave gate kernels park on the aux / side queues for as long as the kernel they wait for runs: they are not work, and they are left out
# of every total and percentage taken from a summary
ROCPROF_NOT_WORK = ("k_gate_wait", "k_gate_set", "k_pipe_probe")
ROCPROF_TAG = {"askubuntu": "askubuntu", "c4": "c4_3200users", "ml20m": "ml20m_3200users", "custom:25024": "mid25k"}


def rocprof_summary(workload, d_precision="fp32"):
    """the newest tracked rocprofv3 kernel summary of this workload (profiles/r<N>_<tag>_kernel_stats.csv): {kernel: (calls, total_ns)}, file name"""
    import csv
    import glob
    import re
    tag = ROCPROF_TAG.get(workload)
    if tag is None:
        return None, None
    if workload == "askubuntu" and d_precision == "fp8":
        tag = "askubuntu_wide_fp8"
    best = None
    for fn in glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % tag)):
        m = re.match(r"r(\d+)([a-z]?)_%s_kernel_stats\.csv$" % re.escape(tag), os.path.basename(fn))
        if m and (best is None or (int(m.group(1)), m.group(2)) > best[0]):
            best = ((int(m.group(1)), m.group(2)), fn)
    if best is None:
        return None, None
    rows = {}
    with open(best[1]) as f:
        for r in csv.DictReader(f):
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
            c, t = rows.get(name, (0, 0.0))
            rows[name] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    return rows, os.path.basename(best[1])


class KernelProfiler:
    """Times individual kernels of the D / G step with HIP events recorded inside the library
    (ltg_probe) and prices them against the roofline with ALGORITHMIC flops / bytes."""

    D_KERNELS = ["d_l1", "d_l2", "d_bwd1", "d_bwd2", "d_adam"]
    # (a kernel that the active code path does not launch records no events and is skipped)
    G_KERNELS = ["enc0_fwd", "enc1", "dec0", "dec1_fwd", "row_dlogits", "dh2", "dec1_bwd_adam", "dz", "wgrad_p0", "dh1", "wgrad_q1",
                 "enc0_grad", "enc0_bwd_adam", "g_tail"]

    def __init__(self, eng, tr, data, args):
        from ltgan import _cabi as cabi
        from ltgan._hip import EventPair
        self.cabi, self.EventPair = cabi, EventPair
        self.eng, self.tr, self.data, self.a = eng, tr, data, args
        # the kernels the active code path launches (csrc/ltg_kernels.hip: fast_on / small_fast / mid_fast)
        fast = (args.variant & 262144) == 0
        small = fast and not eng.sharded and eng.I <= 4096 and eng.I % 4 == 0
        if small:
            self.G_KERNELS = ["enc0_fwd", "enc1", "dec0", "dec1_fwd", "row_dlogits", "dh2", "dz", "dh1", "g_tail"]
        elif fast:
            self.G_KERNELS = ["enc0_fwd", "enc1", "dec0", "dec1_fwd", "dh2", "dec1_bwd_adam", "dz", "dh1", "enc0_grad", "g_tail"]
            if eng.I >= 8192:
                self.G_KERNELS.append("enc0_bwd_adam")     # its own launch(es): the dense sweep of W_q0, or the lazy clock's two kernels
        else:
            self.G_KERNELS = ["enc0_fwd", "enc1", "dec0", "dec1_fwd", "dh2", "dec1_bwd_adam", "dz", "wgrad_p0", "dh1", "wgrad_q1", "enc0_bwd_adam"]
        self.samples = []        # (event pair, n_pairs, nnz) recorded in the timed region
        self.extra_samples = {}  # probe id -> samples of the extra ids of hook()
        self.pool = []           # event pairs created BEFORE the timed region (reserve())

    def reserve(self, n):
        while len(self.pool) < n:
            self.pool.append(self.EventPair())

    def _probe(self, name):
        ev = self.pool.pop() if self.pool else self.EventPair()
        p = self.cabi.ltg_probe(self.cabi.KERNEL_IDS[name], 0, ev.start, ev.stop)
        return ev, p

    def _shapes(self, b):
        v = self.data.view(b)
        nnz = int(self.data.idx.train.indptr[v["hi"]] - self.data.idx.train.indptr[v["lo"]])
        return dict(B=v["hi"] - v["lo"], n_real=v["n_real"], n_fake=v["n_slots"], nnz=nnz)

    def work(self, name, sh):
        """algorithmic (flops, bytes) of one launch; DESIGN.md section 'Kernels' has the derivations"""
        e = self.eng
        B, I, H, Z = sh["B"], e.I, e.H, e.Z
        h0, h1, h2, h3 = e.h0, e.h1, e.h2, e.h3
        h12 = h1 + h2
        n = sh["n"]
        P = h0 * h1 + h1 + h0 * h2 + h2 + h12 * h3 + h3 + h3 + 1
        ks = (n + 255) // 256
        d_forked = bool(getattr(e, "_dfork", None) is not None and e._dfork.ok)
        stream = self.a.precision == "bf16" and I >= 8192 and I % 8 == 0 and B <= 128     # stream_ok() of csrc/ltg_kernels.hip
        if stream:
            kchunk = max(128, -(-(-(-I // 256)) // 32) * 32)  # dh2_stream_chunk()
        else:
            kchunk = max(256, -(-(-(-I // 64)) // 32) * 32)  # dh2_kchunk()
        nsplit = -(-I // kchunk)
        wread = 2 * 608 * I if stream else 4 * I * H        # bf16 shadow of W_p1t vs fp32 rows converted on the fly
        shadow_w = 2 * 608 * I if stream else 0             # the Adam epilogue refreshes the shadow
        w = {
            "enc0_fwd": (2 * sh["nnz"] * H, 4 * (sh["nnz"] * H + B * H)),
            "enc1": (2 * B * H * 2 * Z, 4 * (B * H + H * 2 * Z + B * 2 * Z)),
            "dec0": (2 * B * Z * H, 4 * (B * Z + Z * H + B * H)),
            "dec1_fwd": (2 * B * H * I, wread + 4 * (I + B * H + B * I)),
            "d_l1": (2 * n * h0 * h12, 4 * (2 * n * h0 + h0 * h12 + n * h12)),
            "d_l2": (2 * n * h12 * h3, 4 * (n * h12 + h12 * h3 + n * h3)),
            # (with the step's fork -- Engine.d_fork, ltg_d_opts.aux_stream -- the launch on the step's own stream is job A alone:
            # dpre1 = ((ds G3) . w3^T) * dact(A1); jobs B / C run in a second launch of the same kernel on the aux stream)
            "d_bwd1": ((2 * n * h3 * h12, 4 * (n * h3 + h12 * h3 + 2 * n * h12)) if d_forked else
                       (2 * n * h3 * h12 + 2 * n * (h12 + 1) * h3 + 2 * n * h3,
                        4 * (n * h3 + h12 * h3 + 2 * n * h12 + n * h3 + ks * (h12 * h3 + 2 * h3 + 1)))),
            "d_bwd2": (2 * n * (h0 + 1) * h12, 4 * (2 * n * h0 + n * h12 + ks * ((h0 + 1) * h12))),
            "d_adam": (0, P * (4 * ks + 24)),
            "dh2": (2 * B * I * H, wread + 4 * (B * I + nsplit * B * H)),
            "dec1_bwd_adam": (2 * I * (H + 1) * B, 24 * (I * H + I) + shadow_w + 4 * (B * I + B * H)),
            "dz": (2 * B * H * Z, 4 * (B * H + Z * H + 2 * B * 2 * Z)),
            "wgrad_p0": (2 * B * (Z + 1) * H, 24 * (Z + 1) * H + 4 * (B * Z + B * H)),
            "dh1": (2 * B * 2 * Z * H, 4 * (B * 2 * Z + H * 2 * Z + 2 * B * H)),
            "wgrad_q1": (2 * B * (H + 1) * 2 * Z, 24 * (H + 1) * 2 * Z + 4 * (B * H + B * 2 * Z)),
            # dense: every row of W_q0 every step; lazy clock: the batch's rows + one row in q0_period + the gradient rows
            "enc0_bwd_adam": (2 * sh["nnz"] * H, (24 * (min(I, sh["nnz"]) + 1 + I // e.q0_period) * H + 4 * min(I, sh["nnz"]) * H) if e.lazy_q0
                              else (24 * (I + 1) * H + 4 * B * H)),
            "row_dlogits": (0, 8 * B * I),
            "enc0_grad": (2 * sh["nnz"] * H, 4 * (sh["nnz"] * H + min(I, sh["nnz"]) * H)),
            # one launch = every Adam update of the generator: 24 B per parameter + the operands of the three gradient products
            "g_tail": (2 * B * ((2 * I if I <= 4096 else 0) * (H + 1) + (Z + 1) * H + (H + 1) * 2 * Z),
                       # (item slabs above 4 096: the item-sized tensors are updated by dec1_bwd_adam / enc0_bwd_adam / enc0_grad, not by the tail)
                       24 * (((2 * H + 1) * I if I <= 4096 else 0) + H * 2 * Z + Z * H + 2 * H + 2 * Z) + 4 * B * ((2 * I if I <= 4096 else 0) + 3 * H + 3 * Z)),
        }
        return w[name]

    def _run_probed(self, name, b, probe):
        tr, d, eng = self.tr, self.data, self.eng
        v = d.view(b)
        if name in self.D_KERNELS:
            eng.d_step(v["real"], v["fake"], keep_prob=tr.d_keep, rng_step=tr._step(), probe=probe)
            n = v["n_real"] + v["n_slots"]
        else:
            eng.g_step(v["batch"], v["fake"], tr.acts, d.fake_cnt[b:], anneal=tr.anneal(), gan_lambda=tr.lam, rng_step=tr._step(),
                       d_rng_step=tr._step(), probe=probe)
            n = v["n_slots"]
        return n

    def calibrate(self, reps=6):
        import torch
        out = {}
        act = self.tr.active or [0]
        nb, S = len(act), self.tr.S
        for name in self.D_KERNELS + self.G_KERNELS:
            evs = []
            for r in range(reps):
                b = act[(7 * r + 3) % nb]
                ev, p = self._probe(name)
                n = self._run_probed(name, b, p)
                sh = self._shapes(b)
                sh["n"] = n
                evs.append((ev, sh, p))
            torch.cuda.synchronize()
            ms = [e.elapsed_ms() for e, _, _ in evs]
            ms = [m for m in ms if m is not None and m > 0]
            if not ms:
                continue
            fl = np.mean([self.work(name, sh)[0] for _, sh, _ in evs])
            by = np.mean([self.work(name, sh)[1] for _, sh, _ in evs])
            # (the fake tower of the G steps: inside every step, or -- Trainer.batched_tower -- a few large launches per phase)
            tower = 1 if getattr(self.tr, "batched_tower", False) else 2
            launches = S * nb * (tower if name in ("d_l1", "d_l2") else 1) + (nb if name in ("enc0_fwd", "enc1", "dec0", "dec1_fwd") else 0)
            out[name] = dict(avg_ms=float(np.mean(ms)), flops=float(fl), bytes=float(by), epoch_ms=float(np.mean(ms)) * launches)
        return out

    def hook(self, name, extra=()):
        """called by the trainer for every D / G step of the timed region; probes every 32nd one (an event pair costs the stream a few
        microseconds of bubbles).  extra: further G-step probe ids (the in-stream exchanges of the one-call sharded step) that take
        turns with `name` on those steps."""
        state = {"i": 0, "turn": 0}
        names = [name] + list(extra)

        def h(kind, b):
            if (kind == "d") != (name in self.D_KERNELS):
                return None
            state["i"] += 1
            if (state["i"] - 1) % 32 or not self.pool:      # the first one, then every 32nd
                return None
            nm = names[state["turn"] % len(names)]
            state["turn"] += 1
            ev, p = self._probe(nm)
            sh = self._shapes(b)
            v = self.data.view(b)
            sh["n"] = (v["n_real"] + v["n_slots"]) if kind == "d" else v["n_slots"]
            (self.samples if nm == name else self.extra_samples.setdefault(nm, [])).append((ev, sh, p))
            return p
        return h

    def extra_us(self):
        """average in-stream duration (us) of the extra probe ids"""
        out = {}
        for nm, smp in self.extra_samples.items():
            ms = [e.elapsed_ms() for e, _, _ in smp]
            ms = [m for m in ms if m is not None]
            if ms:
                out[nm] = {"avg_us": float(np.mean(ms)) * 1e3, "samples": len(ms)}
        return out

    def table(self, calib):
        out = {}
        for name, c in calib.items():
            bf = self.a.precision == "bf16" and name in ("dec1_fwd", "dh2", "dec1_bwd_adam")
            d_low = getattr(self.a, "d_precision", "fp32") in ("fp8", "bf16") and name.startswith("d_") and name != "d_adam"
            peak_f = PEAK["mfma_bf16"] if (bf or d_low) else PEAK["mfma_fp32"]
            avg, fl, by = c["avg_ms"], c["flops"], c["bytes"]
            if avg <= 0:
                continue
            if by / PEAK["hbm"] >= fl / peak_f:
                out[name] = {"bound": "hbm", "frac": round(by / (avg * 1e-3) / PEAK["hbm"], 4), "avg_us": round(avg * 1e3, 2)}
            else:
                out[name] = {"bound": "mfma", "frac": round(fl / (avg * 1e-3) / peak_f, 4), "avg_us": round(avg * 1e3, 2)}
        return out

    def dominant_by_total_time(self, rows, fn):
        """the kernel with the largest calls x average duration in the tracked summary (every launch of a forked kernel counted; the gate kernels
        are not work), priced against its roofline with the algorithmic work of one STEP's launches of it"""
        work = {k: v for k, v in rows.items() if k not in ROCPROF_NOT_WORK}
        if not work:
            return None
        total = sum(t for _, t in work.values())
        kn = max(work, key=lambda k: work[k][1])
        calls, t = work[kn]
        probe = next((p for p, names in ROCPROF_KERNELS.items() if kn in names), None)
        out = {"kernel": kn, "calls": calls, "avg_us": t / calls / 1e3, "share_of_kernel_time": t / total, "summary": fn,
               "note": "share and ranking exclude k_gate_wait / k_gate_set (parked one-wave kernels)"}
        if probe is None or probe == "enc0_bwd_adam":
            return out
        act = self.tr.active or [0]
        e = self.eng
        forked_was = getattr(e, "_dfork", None)
        fl, by = [], []
        for b in act:
            sh = self._shapes(b)
            v = self.data.view(b)
            sh["n"] = (v["n_real"] + v["n_slots"]) if probe in self.D_KERNELS else v["n_slots"]
            if probe == "d_bwd1":      # the step's WHOLE stage 1 (job A + jobs B / C: both launches are in the summary's row)
                n, h12, h3 = sh["n"], e.h1 + e.h2, e.h3
                ks = (n + 255) // 256
                f_, b_ = (2 * n * h3 * h12 + 2 * n * (h12 + 1) * h3 + 2 * n * h3, 4 * (n * h3 + h12 * h3 + 2 * n * h12 + n * h3 + ks * (h12 * h3 + 2 * h3 + 1)))
            else:
                f_, b_ = self.work(probe, sh)
            fl.append(f_)
            by.append(b_)
        fl, by = float(np.mean(fl)), float(np.mean(by))
        per_step = 2 if (probe == "d_bwd1" and forked_was is not None and forked_was.ok) else 1      # launches of this kernel per step
        dur = per_step * t / calls * 1e-9
        bf = self.a.precision == "bf16" and probe in ("dec1_fwd", "dh2", "dec1_bwd_adam")
        d_low = getattr(self.a, "d_precision", "fp32") in ("fp8", "bf16") and probe.startswith("d_") and probe != "d_adam"
        peak_f = PEAK["mfma_bf16"] if (bf or d_low) else PEAK["mfma_fp32"]
        if by / PEAK["hbm"] >= fl / peak_f:
            out.update(bound="hbm", achieved=by / dur / 1e9, peak=PEAK["hbm"] / 1e9, unit="GB/s", frac=by / dur / PEAK["hbm"])
        else:
            out.update(bound="mfma", achieved=fl / dur / 1e12, peak=peak_f / 1e12, unit="TFLOP/s", frac=fl / dur / peak_f)
        out["launches_per_step"] = per_step
        if probe.startswith("d_") and not d_low:
            out["d_arith"] = getattr(e, "d_arith", None)     # fp32 FLOPs against the fp32 matrix peak, whichever pipe forms the products
        return out

    def roofline(self, name, calib):
        ms = [e.elapsed_ms() for e, _, _ in self.samples]
        ms = [m for m in ms if m is not None]
        if ms:
            avg = float(np.mean(ms))
            fl = float(np.mean([self.work(name, sh)[0] for _, sh, _ in self.samples]))
            by = float(np.mean([self.work(name, sh)[1] for _, sh, _ in self.samples]))
        elif calib and name in calib:
            avg, fl, by = calib[name]["avg_ms"], calib[name]["flops"], calib[name]["bytes"]
        else:
            return None
        bf = self.a.precision == "bf16" and name in ("dec1_fwd", "dh2", "dec1_bwd_adam")
        # the discriminator's e4m3 GEMMs use the NON-scaled v_mfma_f32_16x16x32_fp8_fp8, which issues at the bf16 rate (MI355X_MICROARCH.md:
        # the ~5 PF figure is the block-scaled form); its bf16 mode likewise
        d_low = getattr(self.a, "d_precision", "fp32") in ("fp8", "bf16") and name.startswith("d_") and name != "d_adam"
        peak_f = PEAK["mfma_bf16"] if (bf or d_low) else PEAK["mfma_fp32"]
        t_f, t_b = fl / peak_f, by / PEAK["hbm"]
        if t_b >= t_f:
            ach, peak, unit, bound = by / (avg * 1e-3) / 1e9, PEAK["hbm"] / 1e9, "GB/s", "hbm"
        else:
            ach, peak, unit, bound = fl / (avg * 1e-3) / 1e12, peak_f / 1e12, "TFLOP/s", "mfma"
        traffic = None
        one_rank = not self.eng.sharded or (self.eng.item_lo == 0 and self.eng.item_hi == self.eng.I_global)     # (world size 1 on the sharded code path: the slab is the table)
        for fn in (() if not one_rank else ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3c_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")):   # (the passes measured the UNSHARDED kernel sizes)   # PMC-derived HBM bytes per launch of this kernel on this workload,
            try:                                                       # measured offline (profiles/README.md); newest round first
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    traffic = json.load(f)["workloads"][self.a.workload][name]["traffic_bytes"]
                break
            except Exception:
                pass
        # the same kernel in the tracked rocprofv3 summary of this workload: `frac` follows from profiles/ (the HIP-event bracket of the live run
        # is ~30 % longer than the kernel at these durations; both are in the line)
        rp_avg_us, rp_file, dom = None, None, None
        rows, rp_file = rocprof_summary(self.a.workload, getattr(self.a, "d_precision", "fp32")) if one_rank else (None, None)
        if rows:
            c, t = 0, 0.0
            for kn in ROCPROF_KERNELS.get(name, ()):
                if kn in rows:
                    c, t = c + rows[kn][0], t + rows[kn][1]
            if c and name != "enc0_bwd_adam":
                rp_avg_us = t / c / 1e3
            dom = self.dominant_by_total_time(rows, rp_file)
        work_per_launch = by if bound == "hbm" else fl
        scale = (1e9 if bound == "hbm" else 1e12)
        ach_rp = (work_per_launch / (rp_avg_us * 1e-6) / scale) if rp_avg_us else None
        ach_ev = ach
        if ach_rp is not None:
            ach = ach_rp
        return {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                "frac_source": ("avg_us_rocprof (%s)" % rp_file) if ach_rp is not None else "avg_us_event_bracket (no tracked rocprofv3 summary of this workload)",
                "avg_us_rocprof": rp_avg_us, "avg_us_event_bracket": avg * 1e3, "achieved_event_bracket": ach_ev, "frac_event_bracket": ach_ev / peak,
                "d_arith": getattr(self.eng, "d_arith", None) if name.startswith("d_") else None,
                "dominant_by_total_time": dom,
                # HBM-bound kernels also against THIS box's device-to-device copy rate (boxes of the pool copy at 5.1-5.5 TB/s of the 8 TB/s spec:
                # a kernel at 0.62 of the spec is at 0.97 of what the box can move)
                "frac_of_copy_ceiling": (ach / self.copy_gbs) if (bound == "hbm" and getattr(self, "copy_gbs", None)) else None, "traffic": traffic,
                "avg_us": avg * 1e3, "launches_timed": len(ms), "algorithmic_flops": fl, "algorithmic_bytes": by,
                "note": "dominant kernel = largest (avg duration x launches per step) among the probed kernels"}


def step_algorithmic_bytes(idx, data, eng, S, active=None, lazy=False):
    """HBM bytes one step (C + S x D + S x G over the workload's batches) has to move with fp32 master weights,
    fp32 Adam moments, sparse X / sparse mask, fused Adam and one write + one read of the [B, I] logits:
      G = 24 P_G + 2 (4 H I) + 4 H min(I, nnz_B) + 8 B I      D = 24 P_D + 8 K h0      C = 4 H I + 4 H min(I, nnz_B) + 4 B I
    with I = the GLOBAL item count (a sharded job moves the same bytes, spread over its GPUs).
    lazy=True: the byte model of the lazy Adam clock of W_q0 instead (engine.lazy_q0): a G step moves the batch's rows of
    W_q0 / m / v and one row in q0_period, not all I of them -- fewer bytes than SURVEY's dense-Adam count."""
    I, H, Z = eng.I_global, eng.H, eng.Z
    h0, h1, h2, h3 = eng.h0, eng.h1, eng.h2, eng.h3
    P_G = (2 * H + 1) * I + H * 2 * Z + Z * H + 2 * H + 2 * Z
    P_D = h0 * h1 + h1 + h0 * h2 + h2 + (h1 + h2) * h3 + h3 + h3 + 1
    indptr = idx.train.indptr
    act = set(range(data.n_batches)) if active is None else set(active)
    tot = 0.0
    for b in range(data.n_batches):
        v = data.view(b)
        B = v["hi"] - v["lo"]
        nnz = int(indptr[v["hi"]] - indptr[v["lo"]])
        K = v["n_real"] + v["n_slots"]
        tot += 4 * H * I + 4 * H * min(I, nnz) + 4 * B * I
        if b in act:
            tot += S * (24 * P_D + 8 * K * h0)
            tot += S * (24 * P_G + 8 * H * I + 4 * H * min(I, nnz) + 8 * B * I)
            if lazy and eng.lazy_q0:
                tot -= S * 24 * H * max(0, I - min(I, nnz) - I // eng.q0_period)
    return tot


def step_fracs(idx, data, eng, S, active, dt, n_gpus):
    """whole-step HBM fraction: the bytes the step really has to move / measured time / (8 TB/s x GPUs).  With the lazy Adam clock of
    W_q0 (item slabs >= 8192) that is the clock's byte model -- the batch's rows and one row in `period` per G step -- so the fraction
    cannot exceed 1; SURVEY 8/d4's dense-Adam count (every row of W_q0 every step: what tf.train.AdamOptimizer itself would move)
    is reported beside it under its own key and CAN exceed 1 -- the clock moves fewer bytes than that model."""
    dense = step_algorithmic_bytes(idx, data, eng, S, active)
    moved = step_algorithmic_bytes(idx, data, eng, S, active, lazy=True)
    pk = PEAK["hbm"] * n_gpus
    return {"step_algorithmic_bytes": moved, "step_frac": moved / dt / pk,
            "step_algorithmic_bytes_dense_adam": dense, "step_frac_vs_dense_adam_bytes": dense / dt / pk}


def copy_ceiling(device, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate of THIS box (read + write bytes / time), the practical HBM ceiling next to the 8 TB/s
    spec the roofline fraction is quoted against: boxes of the pool differ by ~10 % in it, and so do the HBM-bound kernels."""
    import torch
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(reps):
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return {"value": best, "unit": "GB/s", "what": "torch D2D copy of 1 GiB, read + write bytes, best of %d (a practical reference, not a hard ceiling: "
                                                   "a kernel with more bytes in flight per CU than torch's copy can read a few percent above it)" % reps}


def visible_gpus():
    """GPUs this process tree may use, WITHOUT loading the HIP runtime (the self-launching parent must stay a process that never
    touched the GPU): the visibility variables if set, else the KFD topology in sysfs (GPU nodes have a non-zero simd_count)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    n, base = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                n += 1 if int(props.get("simd_count", "0")) > 0 else 0
            except (OSError, ValueError):
                pass
    except OSError:
        return 0
    return n


def choose_backend(world, n_devices, env=None):
    """(backend, note) of a rank: LTGAN_DIST_BACKEND if set (the self-launching parent sets gloo when it SAW fewer GPUs than ranks);
    otherwise nccl (= RCCL) when every rank can have its own GPU and gloo -- a test rig whose ranks share devices and whose exchanges are
    host callbacks, a different system from the one the metric is about -- when not.  The note goes into the result line."""
    env = os.environ if env is None else env
    forced = env.get("LTGAN_DIST_BACKEND")
    if forced:
        return forced, ("LTGAN_DIST_BACKEND=%s" % forced) + (" (test rig: %d ranks share %d GPU(s))" % (world, n_devices) if forced != "nccl" else "")
    if world > n_devices:
        return "gloo", "fallback: %d ranks on %d visible GPU(s) -- ranks share devices, exchanges through host callbacks" % (world, n_devices)
    return "nccl", "one GPU per rank"


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N FRESH rank processes with torch.distributed.run as a
    CHILD of this process (which never loads the HIP runtime: it imports neither torch nor the library), relay their output and
    exit with their code.
    The reference is a single command without a launcher too (train.py:359-381)."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    n_dev = visible_gpus()      # from sysfs / the visibility variables: this parent never loads the HIP runtime
    if 0 < n_dev < a.gpus and "LTGAN_DIST_BACKEND" not in env:
        # fewer GPUs than ranks (single-GPU test box): RCCL cannot put two ranks on one device -> gloo, ranks share GPUs
        env["LTGAN_DIST_BACKEND"] = "gloo"
        print("bench.py: %d rank(s) on %d visible GPU(s): using the gloo backend (ranks share devices)" % (a.gpus, n_dev), file=sys.stderr, flush=True)
    # (n_dev == 0: sysfs unreadable or restricted by the container -- nothing is forced here; every rank decides from
    # torch.cuda.device_count() once it runs, see choose_backend)
    # --rdzv-endpoint 127.0.0.1:0 lets the launcher's own store pick a free port (no bind-then-close race in this process)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--rdzv-backend", "c10d",
           "--rdzv-endpoint", "127.0.0.1:0", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    last_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"') and line.rstrip().endswith("}"):
            if last_json is not None:
                sys.stderr.write(last_json + "\n")     # an earlier result-shaped line: kept visible, never swallowed
            last_json = line.rstrip()       # rank 0's result line: relayed last, alone
        else:
            sys.stdout.write(line)
    rc = proc.wait()
    sys.stdout.flush()
    if rc != 0:
        print("bench.py: a rank of the %d-rank job failed (exit code %d)" % (a.gpus, rc), file=sys.stderr, flush=True)
        raise SystemExit(rc)
    if last_json is None:
        print("bench.py: the %d-rank job printed no result line" % a.gpus, file=sys.stderr, flush=True)
        raise SystemExit(1)
    print(last_json, flush=True)
    raise SystemExit(0)


def pipe_facts(pipe):
    """what the one-call G step did beside its kernels (include/ltg.h, ABI v13): steps that found their rows of W_q0 caught up by the step
    before (no catch-up launch of their own) out of all steps, and whether the weight update wrote a second shadow buffer"""
    if pipe is None:
        return None
    return {"steps": int(pipe.c.seq), "steps_caught_up_ahead": int(pipe.ahead_calls), "second_shadow_buffer": pipe.shadow is not None,
            "tail_stream": pipe.tail_stream is not None}


def warm_moments(eng):
    """--warm-moments: Adam moments of W_q0 as after long training (every item seen at some point)"""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(7)
    eng.g_flush()
    eng.g_m[0].normal_(0.0, 1e-4, generator=g)
    eng.g_v[0].uniform_(1e-9, 1e-7, generator=g)


def other_workloads(a, device, users=6400, copy_gbs=None):
    """BASELINE configs 3 and 4 (item counts 20 000 / 200 000) on ONE GPU, bounded to 64 batches of 100 users each so the
    default run stays short: users/s of C + S x D + S x G over those batches, the dominant kernel's HBM fraction (HIP events
    recorded by the library around that kernel in the timed epochs) and the whole-step fraction."""
    import torch
    from ltgan.engine import Engine
    from ltgan.trainer import Trainer
    out = {}
    for key, name in (("c3", "ml20m"), ("c4", "c4")):
        idx, data, desc = load_workload(name, a.batch_size, device, users)
        eng = Engine(idx.n_items, h_sizes=a.h_sizes, precision=a.precision, d_precision=a.d_precision, d_arith=a.d_arith, device=device)
        eng.cfg.tuning = a.variant
        warm = not a.cold_moments      # configs 3 / 4 are 136 000 / 1 000 000 users: every row of W_q0 carries moments there
        if warm:
            warm_moments(eng)
        tr = Trainer(eng, data, num_sub_epochs=a.sub_epochs)
        tr.epoch()
        aa = argparse.Namespace(**vars(a))
        aa.workload = name
        prof = KernelProfiler(eng, tr, data, aa)
        prof.copy_gbs = copy_gbs
        kname = "dec1_bwd_adam"
        if not a.no_probe:
            prof.reserve(256)
            tr.probe_hook = prof.hook(kname)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ph = [tr.epoch() for _ in range(2)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        tr.probe_hook = None
        r = prof.roofline(kname, None) if not a.no_probe else None
        nb = max(1, len(tr.active))
        out[key] = {"workload": name, "users": data.N, "items": data.I, "batches": data.n_batches, "value": data.N / dt, "unit": "users/s",
                    "ms_per_step": dt * 1e3, "g_step_us": float(np.median([p["t_g"] for p in ph])) / (a.sub_epochs * nb) * 1e6,
                    "d_step_us": float(np.median([p["t_d"] for p in ph])) / (a.sub_epochs * nb) * 1e6,
                    "warm_moments": bool(warm),
                    "lazy_q0": bool(eng.lazy_q0),
                    "handover": getattr(getattr(tr, "pipe", None), "handover", None),     # how the one-call G step's streams meet (engine.py: _pipe_ready)
                    "pipe": pipe_facts(getattr(tr, "pipe", None)),
                    **step_fracs(idx, data, eng, a.sub_epochs, tr.active, dt, 1),
                    "dominant_kernel": None if r is None else {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_copy_ceiling", "avg_us", "traffic")}}
        del tr, prof, eng, data, idx
        torch.cuda.empty_cache()
    return out


def rank_proxy_leg(a, device, users=6400, copy_gbs=None, items=25024):
    """What ONE rank of an 8-GPU item-sharded C4 run does per step, measured on this GPU (north_star's >= 6x strong scaling): the
    25 024-item slab a rank owns (200 000 / 8, 64-item multiples), on the sharded code path -- ShardedTrainer, ltg_g_step_sharded with its three
    exchanges issued in-stream through RCCL's own entry points -- at WORLD SIZE 1, warm moments, the discriminator replicated as it is at
    config.ini's sizes.  The exchanges therefore cost their launch + a one-rank collective; xGMI latency is NOT in these numbers."""
    import torch
    import torch.distributed as dist
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.sharded import ShardedTrainer, item_slab
    name = "custom:%d" % items
    made_group = False
    tr = None
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
            dist.init_process_group("nccl", rank=0, world_size=1)
            made_group = True
        idx, _, desc = load_workload(name, a.batch_size, device, users)
        lo, hi = item_slab(idx.n_items, 0, 1)
        data = DeviceData(idx, a.batch_size, device, item_lo=lo, item_hi=hi)
        eng = Engine(idx.n_items, h_sizes=a.h_sizes, precision=a.precision, d_precision=a.d_precision, d_arith=a.d_arith, device=device, item_lo=lo, item_hi=hi)
        eng.cfg.tuning = a.variant
        warm_moments(eng)
        tr = ShardedTrainer(eng, data, num_sub_epochs=a.sub_epochs, d_split=None)
        tr.epoch()
        aa = argparse.Namespace(**vars(a))
        aa.workload = name
        prof = KernelProfiler(eng, tr, data, aa)
        prof.copy_gbs = copy_gbs
        one_call = getattr(tr, "pipe", None) is not None and getattr(tr, "comm", None) is not None
        if not a.no_probe and one_call:
            prof.reserve(256)
            tr.probe_hook = prof.hook("dec1_bwd_adam", ("exch_h1", "exch_rowpart", "exch_dh2"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ph = [tr.epoch() for _ in range(2)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        tr.probe_hook = None
        nb = max(1, len(tr.active))
        r = prof.roofline("dec1_bwd_adam", None) if (not a.no_probe and one_call) else None
        out = {"workload": name, "users": data.N, "items": data.I, "batches": data.n_batches, "value": data.N / dt, "unit": "users/s",
               "g_step_us": float(np.median([p["t_g"] for p in ph])) / (a.sub_epochs * nb) * 1e6,
               "d_step_us": float(np.median([p["t_d"] for p in ph])) / (a.sub_epochs * nb) * 1e6,
               "exchanges_us": {k: round(v["avg_us"], 2) for k, v in prof.extra_us().items()},
               "one_call": bool(one_call), "transport": getattr(getattr(tr, "comm", None), "kind", None),
               "rccl_ranks": getattr(getattr(tr, "comm", None), "count", None), "warm_moments": True,
               "handover": getattr(getattr(tr, "pipe", None), "handover", None),
               **step_fracs(idx, data, eng, a.sub_epochs, tr.active, dt, 1),
               "dominant_kernel": None if r is None else {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_copy_ceiling", "avg_us", "traffic")},
               "note": "per-rank proxy of an 8-GPU item-sharded C4 run: world size 1, exchanges in-stream on RCCL, no xGMI latency"}
        if copy_gbs:
            out["step_frac_of_copy_ceiling"] = out["step_frac"] * PEAK["hbm"] / 1e9 / copy_gbs
        return out
    except Exception as e:      # (a box whose RCCL cannot come up must not cost the headline line)
        if tr is not None and hasattr(tr, "abort"):
            tr.abort()
            tr = None
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        if tr is not None:
            tr.close()
        if made_group and dist.is_initialized():
            dist.destroy_process_group()
        torch.cuda.empty_cache()


def main():
    a = parse()
    a.h_sizes = tuple(int(x) for x in a.d_sizes.split(","))
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        self_launch(a)                       # never returns
    if env_world is not None and int(env_world) != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s (run `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)" % (a.gpus, env_world))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    backend, backend_note = choose_backend(world, torch.cuda.device_count())      # "nccl" is RCCL on ROCm
    local %= max(1, torch.cuda.device_count())      # more ranks than GPUs only on the gloo test rig
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    from ltgan.engine import Engine
    from ltgan.trainer import Trainer
    workload = a.workload or ("askubuntu" if world == 1 else "c4")
    a.workload = workload
    # every synthetic workload stands for a configuration whose users cover the whole item table (136 000 / 1 000 000 users): W_q0
    # carries Adam moments in every row there, so that is what is timed unless --cold-moments asks for the bounded sample's own state
    a.warm = (a.warm_moments or workload != "askubuntu") and not a.cold_moments
    mode = a.parallelism or ("item-shard" if world > 1 else "single")
    if mode == "item-shard" and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=0, world_size=1)      # exercises the RCCL code path on one GPU
    idx, data, desc = load_workload(workload, a.batch_size, device, a.users)
    n1_ref = None
    if mode == "item-shard":
        from ltgan.dataset import DeviceData
        from ltgan.sharded import ShardedTrainer, item_slab
        if world > 1 and rank == 0:
            # same workload on ONE GPU, measured in this very job (the other ranks wait): the strong-scaling reference
            e1 = Engine(idx.n_items, h_sizes=a.h_sizes, precision=a.precision, d_precision=a.d_precision, d_arith=a.d_arith, device=device)
            if a.warm:
                warm_moments(e1)
            t1 = Trainer(e1, data, num_sub_epochs=a.sub_epochs)
            t1.epoch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t1.epoch()
            torch.cuda.synchronize()
            n1_ref = data.N / (time.perf_counter() - t0)
            del e1, t1
            torch.cuda.empty_cache()
        lo, hi = item_slab(idx.n_items, rank, world)
        data = DeviceData(idx, a.batch_size, device, item_lo=lo, item_hi=hi)
        eng = Engine(idx.n_items, h_sizes=a.h_sizes, precision=a.precision, d_precision=a.d_precision, d_arith=a.d_arith, device=device, item_lo=lo, item_hi=hi)
        eng.cfg.tuning = a.variant
        if a.warm:
            warm_moments(eng)
        tr = ShardedTrainer(eng, data, num_sub_epochs=a.sub_epochs, d_split={"auto": None, "on": True, "off": False}[a.d_split])
    else:
        eng = Engine(idx.n_items, h_sizes=a.h_sizes, precision=a.precision, d_precision=a.d_precision, d_arith=a.d_arith, device=device)
        eng.cfg.tuning = a.variant
        if a.warm:
            warm_moments(eng)
        tr = Trainer(eng, data, num_sub_epochs=a.sub_epochs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # (an exception between here and close() must not leave this rank's RCCL communicator alive while its peers sit in an in-stream collective:
    # give it up without waiting -- ShardedTrainer.abort = ncclCommAbort -- and re-raise)
    try:
        for _ in range(a.warmup):
            tr.epoch()
        # ---- live per-kernel timing (HIP events recorded by the library around ONE kernel per call)
        prof = KernelProfiler(eng, tr, data, a)
        if a.no_probe:
            calib, dominant = None, None
        elif mode == "item-shard":
            calib, dominant = None, "dec1_bwd_adam"        # monolithic probe steps would desynchronise the shards
        else:
            calib = prof.calibrate() if rank == 0 else None
            dominant = max(calib, key=lambda k: calib[k]["epoch_ms"]) if calib else None
        one_call = getattr(tr, "pipe", None) is not None and getattr(tr, "comm", None) is not None
        exch = ("exch_h1", "exch_rowpart", "exch_dh2") if (mode == "item-shard" and one_call) else ()
        tr.probe_hook = prof.hook(dominant, exch) if (dominant and rank == 0) else None
        if tr.probe_hook:
            prof.reserve(min(2048, a.steps * a.sub_epochs * data.n_batches // 32 + 1))
        # who really runs this job: one line per rank (device identity as the runtime reports it), world size as RCCL reports it
        ranks_info = None
        if dist.is_initialized():
            pr_ = torch.cuda.get_device_properties(local)
            me = {"rank": rank, "local_rank": local, "pid": os.getpid(), "gpu_uuid": str(getattr(pr_, "uuid", "")), "gpu": pr_.name,
                  "pci_bus_id": getattr(pr_, "pci_bus_id", None), "cus": pr_.multi_processor_count}
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, me)
        barrier()
        t0 = time.perf_counter()
        phases = []
        for _ in range(a.steps):
            phases.append(tr.epoch())
        barrier()
        dt = time.perf_counter() - t0
        tr.probe_hook = None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        replicas = mode == "replicas" and world > 1
        data_users = data.N
        users = data.N * a.steps * (world if replicas else 1)   # replicas: every rank processes the full workload
        value = users / dt
        res = {
            "metric": "train users/sec at BATCH_SIZE=%d" % a.batch_size,
            "value": value, "unit": "users/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak" if replicas else "strong", "vs_baseline": None,
            "dtype": a.precision, "data": desc,
            "config": {"workload": a.workload, "users": data.N, "items": data.I, "batches": data.n_batches,
                       "sub_epochs": a.sub_epochs, "batch_size": a.batch_size, "warm_moments": bool(a.warm),
                       "d_arith": eng.d_arith if a.d_precision == "fp32" else a.d_precision,
                       "backend": ("%s (%s)" % (backend, "RCCL over xGMI" if backend == "nccl" else "test rig: ranks share GPUs")) if dist.is_initialized() else "none",
                       "backend_choice": backend_note if dist.is_initialized() else None,
                       "parallelism": ("item-shard x%d (RCCL: 2 all-reduce [B,600] + 1 all-gather [B,5] per G step; D step %s)" %
                                       (world, "pair rows split + gradient all-reduce" if getattr(tr, "d_split", False) else "replicated"))
                       if mode == "item-shard" else ("replicas x%d" % world if replicas else "single GPU")},
            "phases_ms": {k: float(np.median([p[k] for p in phases]) * 1e3) for k in ("t_create", "t_d", "t_g")},
        }
        if mode == "item-shard":
            comm = getattr(tr, "comm", None)
            res["sharded_step"] = {
                "one_call": bool(one_call),                          # ltg_g_step_sharded: every launch and the three exchanges from one C call
                "handover": getattr(getattr(tr, "pipe", None), "handover", None),   # fork / join of the weight update: "device-words" | "events"
                "pipe": pipe_facts(getattr(tr, "pipe", None)),
                "transport": getattr(comm, "kind", "torch.distributed (%s), step cut at its exchange points" % backend),
                "rccl_ranks": getattr(comm, "count", None) if getattr(comm, "kind", "") == "rccl-direct" else None,   # ncclCommCount
                "ranks": ranks_info,
                "distinct_gpus": len({r["gpu_uuid"] or r["pci_bus_id"] or r["local_rank"] for r in ranks_info}) if ranks_info else 1,
            }
    except BaseException:
        if hasattr(tr, "abort"):
            tr.abort()
        raise
    if dist.is_initialized():
        if hasattr(tr, "close"):
            tr.close()                  # the step's own RCCL communicator (ncclCommDestroy) before the group it was created over goes
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if mode == "item-shard":
            ex = prof.extra_us()
            nb_, S_ = max(1, len(tr.active)), a.sub_epochs
            res["sharded_step"]["g_step_us"] = res["phases_ms"]["t_g"] * 1e3 / (nb_ * S_)
            res["sharded_step"]["d_step_us"] = res["phases_ms"]["t_d"] * 1e3 / (nb_ * S_)
            res["sharded_step"]["exchanges_us"] = {k: round(v["avg_us"], 2) for k, v in ex.items()}
            res["sharded_step"]["exchanges_note"] = ("HIP events on the step's stream around each in-stream collective call (every 32nd G step, rank 0): "
                                                     "time the step's critical path spends in the exchange, queueing behind slower ranks included")
        cc = copy_ceiling(device)
        prof.copy_gbs = cc["value"]
        res["roofline"] = prof.roofline(dominant, calib) if dominant else None
        if res["roofline"] is None:
            res["roofline"] = {"kernel": None, "note": "kernel probes disabled (--no-probe)" if a.no_probe else "no probed launch fell into the timed region"}
        # whole-step fraction: algorithmic bytes of one step / measured step time / (8 TB/s x GPUs) (step_fracs: never above 1)
        res["roofline"].update(step_fracs(idx, data, eng, a.sub_epochs, tr.active, dt / a.steps, 1 if replicas else world))
        if eng.lazy_q0:
            res["roofline"]["lazy_q0"] = {"period": eng.q0_period, "warm_moments": bool(a.warm)}
        res["roofline"]["copy_ceiling_this_box"] = cc
        res["roofline"]["step_frac_of_copy_ceiling"] = res["roofline"]["step_frac"] * PEAK["hbm"] / 1e9 / cc["value"]
        if calib:
            res["kernels_us"] = {k: round(v["avg_ms"] * 1e3, 2) for k, v in calib.items()}
            # every probed kernel against ITS roofline (calibration pass: HIP events around one kernel per call, same brackets as `roofline`):
            # the dominant kernel changes as kernels get faster (round 5: fk_g_tail -> fk_d_l2), the table keeps them comparable
            res["kernels_roofline"] = prof.table(calib)
        if n1_ref:
            res["n1_same_workload"] = {"value": n1_ref, "unit": "users/s", "note": "same workload, unsharded, on rank 0's GPU in this job"}
            res["strong_scaling_vs_1gpu"] = value / n1_ref
        if world == 1 and workload == "askubuntu" and not a.no_other_workloads:
            del tr, prof, eng, data
            torch.cuda.empty_cache()
            res["other_workloads"] = other_workloads(a, device, copy_gbs=cc["value"])
            # the scaling bound, driver-observed: what one rank of an 8-GPU C4 run does per step against the one-GPU step of the same table
            rp = rank_proxy_leg(a, device, copy_gbs=cc["value"])
            res["other_workloads"]["rank_proxy"] = rp
            c4 = res["other_workloads"].get("c4")
            if c4 and "g_step_us" in rp:
                res["projected_strong_scaling_8gpu_upper_bound"] = (c4["g_step_us"] + c4["d_step_us"]) / (rp["g_step_us"] + rp["d_step_us"])
                res["projected_strong_scaling_note"] = ("(c4.g_step_us + c4.d_step_us) / (rank_proxy.g_step_us + rank_proxy.d_step_us): one GPU's step over the "
                                                        "200 000-item table against one rank's step over its 25 024-item slab, both measured in this run; "
                                                        "xGMI latency of the three exchanges is NOT in it (world size 1) -- an upper bound, not a measurement")
        # ONE workload across every N: the headline `value` is Askubuntu_Sample at N = 1 (BASELINE's metric configuration) and the
        # C4-shaped synthetic at N > 1, so a 1 -> 8 curve is read from this key -- the C4-shaped bounded sample (200 000 items,
        # 6 400 users unless --users) at this run's N
        c4v = None
        if workload == "c4":
            c4v = value
        elif "other_workloads" in res and "c4" in res["other_workloads"]:
            c4v = res["other_workloads"]["c4"]["value"]
        if c4v is not None:
            res["c4_same_workload"] = {"value": c4v, "unit": "users/s", "n_gpus": world, "workload": "c4 (200000 items; %d users)" %
                                       (data_users if workload == "c4" else 6400)}
        if not a.no_cpu_baseline and world == 1:
            from oracle.cpu_port import time_cpu_baseline   # oracle/ is only ever the baseline / checker
            res["cpu_baseline"] = time_cpu_baseline(idx, budget_s=a.cpu_seconds, S=a.sub_epochs, batch_size=a.batch_size)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL prints its banner through C stdio: get it out BEFORE the JSON line
        except Exception:
            pass
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
