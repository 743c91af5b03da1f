"""Does the G step of a 200 000-item slab depend on WHERE its tables were allocated?  The full C4 runs of round 4 came out in two modes
(weight update 571 vs 615-621 us, same box, same build, process after process).  One process, the engine built and dropped several times:

    python scripts/placement_probe.py [trials] [items] [mode]

mode "plain": tensors allocated as the engine does; "first": a 6-GiB block allocated and released first (the caching allocator then serves
the tables out of ONE segment)."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                   # noqa: E402
from ltgan.dataset import DeviceData           # noqa: E402
from ltgan.engine import Engine                # noqa: E402
from ltgan.synthetic import synthetic_index    # noqa: E402
from ltgan.trainer import Trainer              # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 5
items = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
idx, _ = synthetic_index("custom:%d" % items, users=1600, seed=21)
for trial in range(trials):
    if mode == "first":
        blk = torch.empty(6 << 30, dtype=torch.uint8, device="cuda:0")
        del blk
    eng = Engine(idx.n_items, lr=1e-4, precision="bf16", seed=5, d_seed=9)
    data = DeviceData(idx, 100, eng.device)
    bench.warm_moments(eng)
    tr = Trainer(eng, data, num_sub_epochs=4, shuffle_seed=4)
    tr.create_phase()
    tr.d_phase()
    tr.g_phase()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.g_phase()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = 4 * len(tr.order)
    print("trial %d (%s): G step %.1f us  [theta at %#x, m at %#x, v at %#x, shadow at %#x]" %
          (trial, mode, dt / n * 1e6, eng.g_p[3].data_ptr(), eng.g_m[3].data_ptr(), eng.g_v[3].data_ptr(), eng.g_shadow.data_ptr()), flush=True)
    del tr, data, eng
    gc.collect()
    torch.cuda.empty_cache()
