"""Soak of the one-call G step's device-word hand-over: the same loop with and without the pipelined step, bit for bit, over many steps
(tests/test_gpu_trajectory.py does two epochs at 9 000 items; this is the long form).  usage: python scripts/soak_onecall.py [items] [epochs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ltgan.dataset import DeviceData          # noqa: E402
from ltgan.engine import Engine               # noqa: E402
from ltgan.synthetic import synthetic_index   # noqa: E402
from ltgan.trainer import Trainer             # noqa: E402

items = int(sys.argv[1]) if len(sys.argv) > 1 else 25024
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
idx, _ = synthetic_index("custom:%d" % items, users=1000, seed=21)
runs = []
for pipe in (False, True):
    eng = Engine(idx.n_items, lr=1e-3, precision="bf16", seed=5, d_seed=9)
    data = DeviceData(idx, 100, eng.device)
    tr = Trainer(eng, data, num_sub_epochs=10, shuffle_seed=4, pipe_step=pipe)
    for _ in range(epochs):
        tr.epoch()
    torch.cuda.synchronize()
    tr.check_pipe()
    if pipe:
        print("hand-over:", tr.pipe.handover, "expired waits:", tr.pipe.expired_waits(), "steps:", tr.pipe.c.seq)
        print("steps without a catch-up launch of their own:", tr.pipe.ahead_calls, "second shadow buffer:", tr.pipe.shadow is not None)
    runs.append([t.clone() for t in eng.g_p + eng.g_m + eng.g_v + eng.d_p])
bad = [k for k, (x, y) in enumerate(zip(*runs)) if not torch.equal(x, y)]
print("tensors:", len(runs[0]), "differing:", bad)
sys.exit(1 if bad else 0)
