"""Worker of tests/test_gpu_sharded.py::test_one_shot_exchange_equals_the_rank_ordered_host_transport: N gloo ranks sharing cuda:0 (fresh
processes).  Every rank trains its item slab twice from equal states -- once with the library's one-shot exchange over HIP-IPC-mapped staging
buffers (ltgan._rccl.OneShotComm, csrc/ltg_oneshot.h), once with host callbacks that add the ranks' contributions in rank order -- and the two
runs must agree BIT FOR BIT: fake pairs, losses, every generator tensor and moment."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.sharded import ShardedTrainer, item_slab
    from ltgan.synthetic import synthetic_index
    workload, users = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0"
    torch.cuda.set_device(dev)
    idx, _ = synthetic_index(workload, users=users, seed=5)
    I = idx.n_items
    lo, hi = item_slab(I, rank, world)
    if len(sys.argv) > 3 and sys.argv[3] == "expiry":
        # a peer that never sends: rank 1 sits out the G phase, so every exchange of rank 0 waits its bound (LTGAN_ONESHOT_LIMIT_MS) and gives up --
        # a counter, not a hang -- and the phase must end in an error, not in silently wrong weights
        from ltgan._cabi import LtgError
        eng = Engine(I, h_sizes=(16, 24, 40, 32), lr=1e-3, precision="bf16", seed=77, d_seed=3, device=dev, item_lo=lo, item_hi=hi)
        data = DeviceData(idx, 100, dev, item_lo=lo, item_hi=hi)
        tr = ShardedTrainer(eng, data, num_sub_epochs=1, shuffle_seed=1, transport="oneshot")
        assert tr.comm is not None and tr.comm.kind == "oneshot-ipc"
        tr.create_phase()
        tr.d_phase()
        torch.cuda.synchronize()
        dist.barrier()
        raised = False
        if rank == 0:
            try:
                tr.g_phase()
            except LtgError as e:
                raised = "gave up" in str(e)
            assert raised, "the G phase of a rank whose peer never sends must raise"
            assert tr.comm.expired_waits() > 0
        dist.barrier()
        tr.close()
        if rank == 0:
            print("ONESHOT_EXPIRY_OK world=%d" % world)
        dist.destroy_process_group()
        return
    runs = []
    for transport in ("oneshot", "host-ordered"):
        eng = Engine(I, h_sizes=(16, 24, 40, 32), lr=1e-3, precision="bf16", seed=77, d_seed=3, device=dev, item_lo=lo, item_hi=hi)
        data = DeviceData(idx, 100, dev, item_lo=lo, item_hi=hi)
        tr = ShardedTrainer(eng, data, num_sub_epochs=2, shuffle_seed=1, transport=transport)
        assert tr.pipe is not None and tr.comm is not None, "the one-call step did not engage"
        assert tr.comm.kind == ("oneshot-ipc" if transport == "oneshot" else "host-ordered (gloo)"), tr.comm.kind
        losses = []
        for _ in range(2):
            tr.create_phase()
            losses.append(tr.d_phase().clone())
            losses.append(tr.g_phase().clone())
        eng.g_flush()
        torch.cuda.synchronize()
        if transport == "oneshot":
            assert tr.comm.expired_waits() == 0, "a device-side wait for a peer's message gave up"
            n_exch = tr.comm.os.seq
        assert tr.pipe.expired_waits() == 0
        runs.append((data.fake_gen.clone(), losses, [t.clone() for t in eng.g_p + eng.g_m + eng.g_v]))
        tr.close()
        dist.barrier()
    a, b = runs
    assert torch.equal(a[0], b[0]), "fake pairs differ"
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y), "losses differ"
    for k, (x, y) in enumerate(zip(a[2], b[2])):
        assert torch.equal(x, y), ("generator tensor / moment differs", k, (x - y).abs().max().item())
    dist.barrier()
    if rank == 0:
        print("ONESHOT_OK world=%d workload=%s exchanges=%d slabs=%s" % (world, workload, n_exch, sorted({b_ - a_ for a_, b_ in (item_slab(I, r, world) for r in range(world))})))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
