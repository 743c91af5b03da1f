"""Upper bound of what hipGraph replay would buy: capture ONE G step and ONE D step of an Askubuntu batch (scalars baked in, so the
replays recompute the same step -- timing only) and compare 500 replays with 500 direct calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ltgan  # noqa
from ltgan.dataset import DeviceData, IndexData
from ltgan.engine import Engine
from ltgan.trainer import Trainer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

idx, data, desc = bench.load_workload("askubuntu", 100, "cuda:0", None)
eng = Engine(idx.n_items, device="cuda:0")
tr = Trainer(eng, data, num_sub_epochs=1)
tr.create_phase()
b = tr.active[3]
v = data.view(b)
N = 500

def g_call():
    eng.g_step(v["batch"], v["fake"], tr.acts, data.fake_cnt[b:], anneal=0.1, rng_step=7, d_rng_step=8, loss_out=tr.g_losses[0])

def d_call():
    eng.d_step(v["real"], v["fake"], keep_prob=0.7, rng_step=9, loss_out=tr.d_losses[0])

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for name, fn in (("G step", g_call), ("D step", d_call)):
    for _ in range(20): fn()
    direct = timed(fn, N)
    eng.adam_t = 5
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    for _ in range(20): g.replay()
    rep = timed(g.replay, N)
    print("%s: direct %.1f us, graph replay %.1f us" % (name, direct, rep))
