"""cProfile of the host side of the item-sharded G phase (world size 1 on RCCL)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1)
from ltgan.dataset import DeviceData
from ltgan.engine import Engine
from ltgan.sharded import ShardedTrainer
from ltgan.synthetic import synthetic_index
idx, _ = synthetic_index("custom:25024", users=3200)
eng = Engine(idx.n_items, device="cuda:0")
data = DeviceData(idx, 100, "cuda:0", item_lo=0, item_hi=idx.n_items)
tr = ShardedTrainer(eng, data, num_sub_epochs=4)
tr.epoch(); tr.create_phase(); tr.d_phase()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
tr.g_phase()
pr.disable(); torch.cuda.synchronize()
n = 4 * len(tr.active)
st = pstats.Stats(pr); st.sort_stats("tottime")
print("steps", n)
st.print_stats(22)
dist.destroy_process_group()
