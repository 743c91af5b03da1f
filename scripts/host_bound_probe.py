"""Host time to ISSUE a G step against the device time of the step, item-sharded trainer at world size 1 on RCCL and the unsharded one.
Short bursts (16 steps after a device sync): a long phase fills the HIP queue and the host is then throttled to the device's pace, which
would report the device time as issue time.  usage: python scripts/host_bound_probe.py [items = 25024]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
from ltgan.dataset import DeviceData
from ltgan.engine import Engine
from ltgan.sharded import ShardedTrainer
from ltgan.trainer import Trainer
from ltgan.synthetic import synthetic_index
items = int(sys.argv[1]) if len(sys.argv) > 1 else 25024
idx, _ = synthetic_index("custom:%d" % items, users=1600)
dev = "cuda:0"
for kind in ("single", "sharded"):
    eng = Engine(idx.n_items, device=dev)
    eng.check_on_flush = False          # (the end-of-phase poison check synchronises the host: it would hide the issue rate this probe is after)
    data = DeviceData(idx, 100, dev, item_lo=0, item_hi=idx.n_items) if kind == "sharded" else DeviceData(idx, 100, dev)
    tr = (ShardedTrainer if kind == "sharded" else Trainer)(eng, data, num_sub_epochs=1)
    for _ in range(3):
        tr.epoch()
    issue, total = [], []
    for _ in range(8):
        tr.create_phase(); tr.d_phase()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.g_phase()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = len(tr.active)
        issue.append((t1 - t0) / n * 1e6); total.append((t2 - t0) / n * 1e6)
    issue.sort(); total.sort()
    print("%s items=%d one_call=%s: issue %.1f us/step, total %.1f us/step (medians of 8 bursts of %d steps; a burst also carries the phase's "
          "tower-ahead launches, the join and the flush)" % (kind, items, tr.pipe is not None, issue[4], total[4], n))
dist.destroy_process_group()
