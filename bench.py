#!/usr/bin/env python3
"""bench.py -- train users/sec at BATCH_SIZE=100 of the Long-Tail-GAN adversarial training path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload askubuntu|ml20m|c4shard]

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
    ap.add_argument("--workload", default="askubuntu")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--sub-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def load_workload(name, batch_size, device):
    from ltgan.dataset import DeviceData, IndexData, materialize_askubuntu
    if name == "askubuntu":
        raw = os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz")
        d = tempfile.mkdtemp(prefix="askubuntu_")
        materialize_askubuntu(raw, d)
        idx = IndexData.from_dir(d)
        desc = "Askubuntu_Sample (10001 users x 1000 items, 179368 interactions; dataset files rebuilt from tests/golden/askubuntu_raw.npz)"
    else:
        from ltgan.synthetic import synthetic_index
        idx, desc = synthetic_index(name)
    return idx, DeviceData(idx, batch_size, device), desc


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from ltgan.engine import Engine
    from ltgan.trainer import Trainer
    idx, data, desc = load_workload(a.workload, a.batch_size, device)
    eng = Engine(idx.n_items, precision=a.precision, device=device)
    tr = Trainer(eng, data, num_sub_epochs=a.sub_epochs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.epoch()
    barrier()
    t0 = time.perf_counter()
    phases = []
    for _ in range(a.steps):
        phases.append(tr.epoch())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    users = data.N * a.steps * world          # replicas: every rank processes the full workload
    value = users / dt
    res = {
        "metric": "train users/sec at BATCH_SIZE=%d" % a.batch_size,
        "value": value, "unit": "users/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": desc,
        "config": {"workload": a.workload, "users": data.N, "items": data.I, "batches": data.n_batches,
                   "sub_epochs": a.sub_epochs, "batch_size": a.batch_size, "parallelism": "replicas x%d" % world},
        "phases_ms": {k: float(np.median([p[k] for p in phases]) * 1e3) for k in ("t_create", "t_d", "t_g")},
    }
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
