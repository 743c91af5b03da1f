"""N>1 path on CPU: world_size-2 gloo run of the item-sharded exchange steps (tests/dist_cpu_worker.py),
plus the host-side sharding of the data."""
import pytest
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_exchange_algebra(world):
    """world 8 = the node the item sharding is built for: eight slabs with an uneven last one, rowpart_all [8][B][5]"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), os.path.join(ROOT, "tests", "dist_cpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "DIST_CPU_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_item_slabs_partition_and_shard_data():
    from ltgan.dataset import DeviceData
    from ltgan.sharded import item_slab
    from ltgan.synthetic import synthetic_index
    for I, R in ((1000, 8), (200000, 8), (20000, 2), (777, 4)):
        slabs = [item_slab(I, r, R) for r in range(R)]
        assert slabs[0][0] == 0 and slabs[-1][1] == I
        assert all(slabs[i][1] == slabs[i + 1][0] for i in range(R - 1)) and all(hi > lo for lo, hi in slabs)
    idx, _ = synthetic_index("custom:500", users=230, seed=3)
    full = DeviceData(idx, 100, "cpu")
    nnz = 0
    for r in range(2):
        lo, hi = item_slab(500, r, 2)
        d = DeviceData(idx, 100, "cpu", item_lo=lo, item_hi=hi)
        assert d.I == hi - lo and d.I_global == 500 and d.n_batches == full.n_batches
        assert np.array_equal(d.row_norm2.numpy(), full.row_norm2.numpy())          # norms are over the FULL row
        loc = d.local_csr
        assert loc.shape == (idx.N, hi - lo) and (loc.indices.max() < hi - lo)
        sub = idx.train[:, lo:hi]
        assert (loc != sub).nnz == 0
        nnz += loc.nnz
        # the per-batch transposed view indexes the LOCAL arrays
        b = 1
        v = d.view(b)
        sl = d.slot[b * d.I:(b + 1) * d.I].numpy()
        e0, e1 = d.ent_off[b], d.ent_off[b + 1]
        u0, u1 = d.uptr_off[b], d.uptr_off[b + 1]
        up = d.uptr[u0:u1].numpy()
        assert up[0] == 0 and up[-1] == e1 - e0 == loc[v["lo"]:v["hi"]].nnz and v["batch"].c.n_unique == len(up) - 1
        pos = d.csr_pos[e0:e1].numpy()
        items = d.indices.numpy()[pos]
        assert np.all(np.diff(items) >= 0)                                                # grouped by local item id
        assert np.array_equal(sl[items], np.repeat(np.arange(len(up) - 1), np.diff(up)))   # slot[] points at the group
        assert (sl >= 0).sum() == len(up) - 1
        assert torch_equal(d.fake_row, full.fake_row) and torch_equal(d.cand_idx, full.cand_idx)   # global lists
    assert nnz == idx.train.nnz


def torch_equal(a, b):
    import torch
    return torch.equal(a, b)


def test_item_slab_rejects_empty_slabs_and_covers_the_items():
    from ltgan.sharded import item_slab
    for n, w in ((1000, 2), (1000, 8), (200000, 8), (100032, 2), (1000, 16)):
        cuts = [item_slab(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(hi > lo for lo, hi in cuts)
    with pytest.raises(ValueError, match="slabs"):
        item_slab(1000, 0, 17)          # ceil(1000/17) -> 64-item multiples leave the last rank with nothing


def test_eval_chunk_is_bounded_by_the_logits_budget():
    from ltgan.trainer import eval_chunk_rows
    assert eval_chunk_rows(1000) >= 20000 and eval_chunk_rows(200000) == (2 << 30) // (4 * 200000) and eval_chunk_rows(10 ** 10) == 1
