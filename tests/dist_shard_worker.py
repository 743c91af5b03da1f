"""Worker of tests/test_gpu_sharded.py: `torchrun --nproc-per-node 2` on ONE GPU (gloo backend, both
ranks on cuda:0) -- runs the item-sharded trainer and, on every rank, the unsharded trainer with the
same seeds, and compares the fake pairs, the loss trajectory and the rank's slab of every tensor."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.sharded import ShardedTrainer, item_slab
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    workload, users, precision = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    mode = sys.argv[4] if len(sys.argv) > 4 else ""
    d_split = mode in ("dsplit", "wide_fp8", "wide_fp8_full")    # pair rows of the discriminator step split over the ranks
    wide = mode in ("wide_fp8", "wide_fp8_full")                 # BASELINE config 5 inside the sharded loop: wide discriminator, fp8 GEMM operands
    if mode == "cutpoints":                                      # the G step cut at its exchange points (torch.distributed collectives)
        os.environ["LTGAN_SHARDED_STEP"] = "0"                   # instead of the one-call step with in-stream exchanges
    # LTGAN_TEST_BACKEND=nccl (tests/test_gpu_sharded.py::test_two_ranks_on_two_gpus_over_rccl, boxes with >= 2 GPUs): one GPU per
    # rank, the exchanges issued in-stream by RCCL itself -- the transport of a real node
    backend = os.environ.get("LTGAN_TEST_BACKEND", "gloo")
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    idx, _ = synthetic_index(workload, users=users, seed=5)
    I = idx.n_items
    hs = (2048, 1024, 512, 256) if mode == "wide_fp8_full" else ((512, 256, 256, 128) if wide else (16, 24, 40, 32))
    dq = "fp8" if wide else "fp32"
    S = 2
    # ---- unsharded reference
    ref = Engine(I, h_sizes=hs, lr=1e-3, precision=precision, seed=77, d_seed=3, device=dev, d_precision=dq)
    ref_tr = Trainer(ref, DeviceData(idx, 100, dev), num_sub_epochs=S, shuffle_seed=1)
    # ---- this rank's shard
    lo, hi = item_slab(I, rank, world)
    eng = Engine(I, h_sizes=hs, lr=1e-3, precision=precision, seed=77, d_seed=3, device=dev, item_lo=lo, item_hi=hi, d_precision=dq)
    data = DeviceData(idx, 100, dev, item_lo=lo, item_hi=hi)
    tr = ShardedTrainer(eng, data, num_sub_epochs=S, shuffle_seed=1, d_split=d_split)
    assert tr.d_split == d_split
    # (every rank takes the same path: one slab below the one-call step's 8 192 items sends ALL ranks onto the cut-point sequence)
    smallest = min(b - a for a, b in (item_slab(I, r, world) for r in range(world)))
    one_call = precision == "bf16" and smallest >= 8192 and mode != "cutpoints"
    assert (tr.pipe is not None) == one_call and (tr.comm is not None) == one_call, (tr.pipe, tr.comm)
    if backend == "nccl" and one_call:
        assert tr.comm.kind == "rccl-direct" and tr.comm.count == world, (tr.comm.kind, tr.comm.count)     # ncclCommCount of the step's own communicator
    p0 = [t.clone() for t in ref.g_p]                            # initial variables: the runs are compared by how far they MOVE
    # ---- ranking metrics over the shards (before training: parameters are identical, only the all-reduce order differs)
    import scipy.sparse as sp
    from ltgan.dataset import EvalData
    from ltgan.sharded import ShardedEvaluator
    from ltgan.trainer import Evaluator
    rs = np.random.default_rng(11)
    n_ev = min(idx.N, 230)
    fold = idx.train[:n_ev]
    te_rows = np.repeat(np.arange(n_ev), 6)
    te = sp.csr_matrix((np.ones(len(te_rows), np.float32), (te_rows, rs.integers(0, I, len(te_rows)))), shape=(n_ev, I))
    te.data[:] = 1.0
    te = te[:, :].tolil()
    te[3] = 0                                                    # a user without held-out items (IDCG == 0: dropped, like the reference)
    te = te.tocsr()
    te.eliminate_zeros()
    m_ref = Evaluator(ref, EvalData(fold, te, dev), chunk=100)
    want = m_ref.run(rng_step=900)
    m_sh = ShardedEvaluator(eng, EvalData(fold, te, dev, item_lo=lo, item_hi=hi), chunk=100)
    got = m_sh.run(rng_step=900)
    a, b = m_ref.out.cpu().numpy(), m_sh.out.cpu().numpy()
    same = np.all(a == b, axis=1).mean()
    assert same > 0.97, ("rows with identical metrics", same)
    assert got["n_users"] == want["n_users"] == n_ev - 1
    for k in ("ndcg", "recall20", "recall50"):
        assert abs(got[k] - want[k]) < 2e-3, (k, got[k], want[k])
    for epoch in range(2):
        ref_tr.create_phase()
        tr.create_phase()
        torch.cuda.synchronize()
        assert torch.equal(data.fake_gen, ref_tr.data.fake_gen), "fake pairs differ (epoch %d)" % epoch
        assert torch.equal(data.fake_pop, ref_tr.data.fake_pop)
        assert torch.equal(data.fake_cnt, ref_tr.data.fake_cnt)
        dl_ref = ref_tr.d_phase().cpu().numpy()[:S, 0]
        dl = tr.d_phase().cpu().numpy()[:S, 0]
        # split: gradient and loss sums meet in another order; fp8: a weight that differs in its last bit can round to the other e4m3 neighbour
        np.testing.assert_allclose(dl, dl_ref, rtol=5e-3 if wide else (2e-5 if d_split else 1e-6))
        gl_ref = ref_tr.g_phase().cpu().numpy()[:S, :6]
        gl = tr.g_phase().cpu().numpy()[:S, :6]
        tol = 2e-5 if precision == "fp32" else 2e-3
        if wide:
            tol = 1e-2
        np.testing.assert_allclose(gl[:, :2], gl_ref[:, :2], rtol=tol)
        np.testing.assert_allclose(gl[:, 2:], gl_ref[:, 2:], rtol=10 * tol, atol=1e-6)
    torch.cuda.synchronize()
    # Generator tensors: 2 epochs x S x (active batches) Adam steps at lr 1e-3 move an element by a few 1e-3 in total, so an absolute
    # bound near that size cannot fail.  Compare the MOVES: mean |difference| against the mean distance travelled, and the share of
    # elements that went somewhere else altogether (an element whose gradient is within rounding of zero takes sign-like Adam
    # steps: it may legitimately go the other way in a run that adds its partial sums in another order).
    eng.g_flush()
    ref.g_flush()
    torch.cuda.synchronize()
    for i in range(8):
        want, init = ref.g_p[i], p0[i]
        if i in (0, 3, 7):
            want, init = want[lo:hi], init[lo:hi]
        mv_ref, mv = (want - init).double(), (eng.g_p[i] - init).double()
        travel = mv_ref.abs().mean().item()       # (rows of W_q0 no user of the run holds never move: the mean is over all of them)
        assert mv_ref.abs().max().item() > 1e-4 and travel > 0, ("the reference run did not move tensor", i)
        diff = (mv - mv_ref).abs()
        rel_mean = diff.mean().item() / travel
        astray = (diff > 0.25 * mv_ref.abs().max().item()).double().mean().item()
        lim = (2e-3, 2e-4) if precision == "fp32" else (3e-2, 3e-3)
        if wide:
            lim = (1e-1, 1e-2)
        assert rel_mean < lim[0] and astray < lim[1], ("gen tensor", i, rel_mean, astray, travel)
    for i in range(8):
        dd = (eng.d_p[i] - ref.d_p[i]).abs()
        if wide:     # four D steps; an element whose gradient is within rounding of zero moves by up to 3.2 lr_t per step in either direction
            assert dd.max().item() < 2e-2 and dd.mean().item() < 1e-4, ("disc tensor", i, dd.max().item(), dd.mean().item())
        else:
            assert dd.max().item() < (2e-5 if d_split else 1e-6), ("disc tensor", i)
    assert eng.adam_t == ref.adam_t
    if tr.pipe is not None:
        assert tr.pipe.expired_waits() == 0 and (ref_tr.pipe is None or ref_tr.pipe.expired_waits() == 0)
    transport = getattr(tr.comm, "kind", "torch.distributed")
    tr.close()
    dist.barrier()
    if rank == 0:
        print("SHARDED_OK world=%d workload=%s precision=%s d_split=%s d_precision=%s backend=%s transport=%s handover=%s path=%s slabs=%s" % (
            world, workload, precision, d_split, dq, backend, transport, getattr(tr.pipe, "handover", None),
            "one-call" if one_call else "cut-points", sorted({b - a for a, b in (item_slab(I, r, world) for r in range(world))})))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
