"""Worker of tests/test_gpu_sharded.py::test_direct_rccl_transport...: ONE rank on RCCL (a single-GPU box cannot host two RCCL ranks).
The item-sharded trainer binds RCCL directly (ltgan._rccl.RcclComm: its own ncclCommInitRank) and hands ncclAllReduce / ncclAllGather
to ltg_g_step_sharded, which issues the three exchanges of every G step in-stream.  With one rank the exchanges are identities, so
the run must equal the unsharded trainer (same one-call step without a communicator) bit for bit -- fake pairs, losses, every tensor."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.sharded import ShardedTrainer
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = "cuda:0"
    idx, _ = synthetic_index("custom:9000", users=330, seed=8)
    runs = []
    for kind in ("unsharded", "rccl"):
        eng = Engine(idx.n_items, h_sizes=(20, 24, 40, 36), lr=1e-3, precision="bf16", seed=5, d_seed=9, device=dev,
                     item_lo=0, item_hi=idx.n_items)
        data = DeviceData(idx, 100, dev, item_lo=0, item_hi=idx.n_items)
        tr = (ShardedTrainer if kind == "rccl" else Trainer)(eng, data, num_sub_epochs=2, shuffle_seed=4)
        assert tr.pipe is not None
        if kind == "rccl":
            assert tr.comm is not None and tr.comm.kind == "rccl-direct" and tr.comm.count == 1, (tr.comm, getattr(tr.comm, "kind", None))
        else:
            assert tr.comm is None
        losses = []
        for _ in range(2):
            tr.create_phase()
            losses.append(tr.d_phase().clone())
            losses.append(tr.g_phase().clone())
        torch.cuda.synchronize()
        runs.append((data.fake_gen.clone(), losses, [t.clone() for t in eng.g_p + eng.g_m + eng.g_v]))
        if kind == "rccl":
            tr.close()
            assert tr.comm is None
    a, b = runs
    assert torch.equal(a[0], b[0]), "fake pairs differ"
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y), "losses differ"
    for k, (x, y) in enumerate(zip(a[2], b[2])):
        assert torch.equal(x, y), ("tensor", k)
    dist.barrier()
    print("RCCL_DIRECT_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
