"""Is the item-sharded G phase host-bound?  World size 1 on RCCL: wall time of issuing the phase (no sync) vs the phase."""
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
items = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
idx, _ = synthetic_index("custom:%d" % items, users=3200)
dev = "cuda:0"
for kind in ("single", "sharded"):
    eng = Engine(idx.n_items, device=dev)
    data = DeviceData(idx, 100, dev, item_lo=0, item_hi=idx.n_items) if kind == "sharded" else DeviceData(idx, 100, dev)
    tr = (ShardedTrainer if kind == "sharded" else Trainer)(eng, data, num_sub_epochs=4)
    tr.epoch()
    tr.create_phase(); tr.d_phase()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.g_phase()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = 4 * len(tr.active)
    print("%s items=%d: issue %.1f us/step, total %.1f us/step" % (kind, items, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
dist.destroy_process_group()
