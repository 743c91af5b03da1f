"""G phase of Askubuntu_Sample under the four combinations of the two aux-stream overlaps (tuning-knob bits 9 and 17)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ltgan  # noqa
from ltgan.engine import Engine
from ltgan.trainer import Trainer
import bench

idx, data, desc = bench.load_workload("askubuntu", 100, "cuda:0", None)
for rep in range(2):
    for name, knob in (("tower+bwd", 0), ("tower only", 131072), ("bwd only", 512), ("none", 512 | 131072)):
        eng = Engine(idx.n_items, device="cuda:0")
        eng.cfg.reserved0 = knob
        tr = Trainer(eng, data, num_sub_epochs=1)
        tr.create_phase()
        tr.d_phase(); tr.g_phase()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.g_phase()
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        print("%-10s G phase %.1f ms (%.1f us / step)" % (name, t * 1e3, t * 1e6 / len(tr.active)))
