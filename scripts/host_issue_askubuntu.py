"""Host issue time vs device time of the D and G phases on Askubuntu_Sample (one GPU)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ltgan.dataset import DeviceData, IndexData, materialize_askubuntu
from ltgan.engine import Engine
from ltgan.trainer import Trainer
d = tempfile.mkdtemp()
materialize_askubuntu(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "askubuntu_raw.npz"), d)
idx = IndexData.from_dir(d)
eng = Engine(idx.n_items, device="cuda:0")
tr = Trainer(eng, DeviceData(idx, 100, "cuda:0"), num_sub_epochs=10)
tr.epoch(); tr.create_phase()
for name, fn in (("D", tr.d_phase), ("G", tr.g_phase)):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = 10 * len(tr.active)
    print("%s phase: issue %.1f us/step, total %.1f us/step" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
