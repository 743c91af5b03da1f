"""Worker of tests/test_dist_cpu.py (CPU, gloo, world_size 2): the item-sharded exchange algebra of
ltgan.sharded (all-reduce of the encoder pre-activation, all-gather + combine of the row partials,
all-reduce of dh2) restated with the oracle's numpy math per item slab, driven through
ShardedTrainer's own collective helpers, must reproduce the unsharded oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import helpers as Hh
    from ltgan.sharded import ShardedTrainer, item_slab
    from oracle import ltg_oracle as O
    dist.init_process_group("gloo")
    rank, R = dist.get_rank(), dist.get_world_size()
    I, B = (300 if R <= 2 else 64 * R - 12), 24         # world size 8: seven slabs of 64 items and an uneven last one (52)
    rng = np.random.default_rng(0)                      # same problem on every rank
    X = Hh.random_history(rng, B, I, mean_nnz=9).toarray().astype(np.float64)
    P = {k: np.asarray(v, np.float64) for k, v in O.init_generator(I, seed=1).items()}
    mask = (rng.random((B, I)) < 0.75).astype(np.float64)
    eps = rng.standard_normal((B, O.Z_DIM))
    S_rows = np.array([0, 0, 3, 7, 7, 20])
    S_cols = np.array([5, 150, 299, 2, 151, 77])
    cnt, sum_y, anneal, lam = 6, 2.5, 0.1, 1.0
    losses, g, F = O.g_loss_and_grads(P, X, mask, 0.75, eps, anneal, lam, S_rows, S_cols, cnt, sum_y)
    lo, hi = item_slab(I, rank, R)
    # a ShardedTrainer shell: only its collective helpers are exercised here (no GPU in this container)
    tr = ShardedTrainer.__new__(ShardedTrainer)
    tr.group, tr.R, tr.rank = None, R, rank
    tr.rowpart = torch.zeros(B * 5)
    tr.rowpart_all = torch.zeros(R * B * 5)
    # stage 1: partial pre-activation over the slab -> all-reduce
    nrm = np.sqrt(np.maximum((X * X).sum(1, keepdims=True), 1e-12))
    h = X / nrm / 0.75 * mask
    pre = torch.from_numpy(h[:, lo:hi] @ P["Wq0"][lo:hi])
    tr._allreduce(pre)
    h1 = np.tanh(pre.numpy() + P["bq0"])
    np.testing.assert_allclose(h1, F["h1"], rtol=1e-12, atol=1e-14)
    # stage 2: replicated middle layers, local logits, row partials -> all-gather
    logits = F["h2"] @ P["Wp1"][:, lo:hi] + P["bp1"][lo:hi]
    m = logits.max(1)
    part = np.zeros((B, 5))
    part[:, 0] = m
    part[:, 1] = np.exp(logits - m[:, None]).sum(1)
    part[:, 2] = (X[:, lo:hi] * logits).sum(1)
    inslab = (S_cols >= lo) & (S_cols < hi)
    np.add.at(part[:, 3], S_rows[inslab], np.exp(logits[S_rows[inslab], S_cols[inslab] - lo] - m[S_rows[inslab]]))
    part[:, 4] = X[:, lo:hi].sum(1)
    tr.rowpart[:] = torch.from_numpy(part.reshape(-1)).float()
    allp = tr._allgather_rowpart(B).double().numpy().reshape(R, B, 5)
    # stage 3: the combine of k_g_combine
    M = allp[:, :, 0].max(0)
    se = (allp[:, :, 1] * np.exp(allp[:, :, 0] - M)).sum(0)
    lse = M + np.log(se)
    nb = allp[:, :, 4].sum(0)
    Pb = (allp[:, :, 3] * np.exp(allp[:, :, 0] - lse)).sum(0)
    negll = (-allp[:, :, 2].sum(0) + nb * lse).mean()
    np.testing.assert_allclose(lse, F["lse"], rtol=1e-6)
    c = lam / cnt * sum_y
    assert abs(negll - F["neg_ll"]) < 1e-5 * abs(F["neg_ll"])
    assert abs(-c * Pb.sum() - losses["gan_loss"]) < 1e-5 * abs(losses["gan_loss"])
    # local dlogits / dh2 partial -> all-reduce
    p = np.exp(logits - lse[:, None])
    ind = np.zeros_like(p)
    ind[S_rows[inslab], S_cols[inslab] - lo] = 1.0
    dlog = (p * nb[:, None] - X[:, lo:hi]) / B - c * p * (ind - Pb[:, None])
    dh2 = torch.from_numpy(dlog @ P["Wp1"][:, lo:hi].T)
    tr._allreduce(dh2)
    np.testing.assert_allclose(dh2.numpy(), F["dh2"], rtol=1e-5, atol=1e-9)
    # the sharded tables need no gradient exchange: the local gradient IS the slab of the full gradient
    np.testing.assert_allclose(F["h2"].T @ dlog, g["Wp1"][:, lo:hi], rtol=1e-5, atol=1e-9)
    # ---- the two entry points ltg_g_step_sharded calls through ltg_comm (include/ltg.h), in their host-callback form: invoked HERE the way
    # the library invokes them -- raw pointers into registered buffers, ncclAllReduce / ncclAllGather argument order, in place
    import ctypes as C
    from ltgan import _cabi as cabi
    from ltgan._rccl import HostComm
    h1pre = torch.full((B, 8), float(rank + 1))
    rp_all = torch.zeros(R * B * 5)
    rp_all[rank * B * 5:(rank + 1) * B * 5] = torch.arange(B * 5, dtype=torch.float32) + 1000 * rank       # this rank's block, written in place
    comm = HostComm(None, [h1pre, rp_all])
    assert comm.c.n_ranks == R and comm.c.rank == rank
    ar = C.cast(comm.c.all_reduce, cabi.ALL_REDUCE_FN)
    ag = C.cast(comm.c.all_gather, cabi.ALL_GATHER_FN)
    n = 5 * 8 + 3                                              # a prefix of the buffer (the last batch of an epoch is short)
    assert ar(h1pre.data_ptr(), h1pre.data_ptr(), n, cabi.LTG_NCCL_FLOAT32, cabi.LTG_NCCL_SUM, None, None) == 0
    flat = h1pre.view(-1)
    assert torch.all(flat[:n] == float(sum(range(1, R + 1)))) and torch.all(flat[n:] == float(rank + 1))
    send = rp_all.data_ptr() + rank * B * 5 * 4
    assert ag(send, rp_all.data_ptr(), B * 5, cabi.LTG_NCCL_FLOAT32, None, None) == 0
    for r in range(R):
        assert torch.equal(rp_all[r * B * 5:(r + 1) * B * 5], torch.arange(B * 5, dtype=torch.float32) + 1000 * r)
    assert ar(12345, 12345, 4, cabi.LTG_NCCL_FLOAT32, cabi.LTG_NCCL_SUM, None, None) != 0       # not a registered buffer: an error code, no exception
    dist.barrier()
    if rank == 0:
        print("DIST_CPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
