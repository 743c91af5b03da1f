"""Three global epochs on the synthetic large-item workloads: every loss finite, generator loss decreasing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ltgan  # noqa
from ltgan.engine import Engine
from ltgan.trainer import Trainer
import bench
for wl, users in (("ml20m", 3200), ("c4", 1600)):
    idx, data, desc = bench.load_workload(wl, 100, "cuda:0", users)
    eng = Engine(idx.n_items, device="cuda:0")
    tr = Trainer(eng, data, num_sub_epochs=10)
    first = last = None
    for ep in range(3):
        tr.create_phase()
        dl = tr.d_phase().cpu().numpy()[:10, 0]
        gl = tr.g_phase().cpu().numpy()[:10, :3]
        assert np.isfinite(dl).all() and np.isfinite(gl).all(), (wl, ep)
        first = gl[0, 1] if first is None else first
        last = gl[-1, 1]
    w = eng.g_p[3]
    assert torch.isfinite(w).all() and torch.isfinite(eng.g_m[3]).all()
    print("%s: vae loss summed over the sub-epoch %.1f -> %.1f, d_loss %.1f, adam_t %d, all finite" % (wl, first, last, dl[-1], eng.adam_t))
