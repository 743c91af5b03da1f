"""Shared builders for the parity tests: seeded problem instances in the oracle's (TF) layout and
the engine's layout, CSR/CSC construction, RNG-derived random tensors."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from oracle import ltg_oracle as O


def random_history(rng, n_rows, n_items, mean_nnz=18, min_nnz=1):
    rows, cols = [], []
    for b in range(n_rows):
        k = int(min(n_items, max(min_nnz, rng.poisson(mean_nnz))))
        it = rng.choice(n_items, size=k, replace=False)
        rows += [b] * k
        cols += sorted(it.tolist())
    X = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_rows, n_items))
    X.sort_indices()
    return X


def csc_view(X):
    """(slot, uptr, rowidx, csr_pos, n_unique) of a CSR matrix: the transposed view ltg_g_step needs."""
    from ltgan.dataset import batch_csc
    X = X.tocsr()
    slot, uptr, rowidx, pos = batch_csc(X, 0, X.shape[0], X.shape[1])
    return slot, uptr, rowidx, pos, len(uptr) - 1


def dropout_mask_dense(seed, step, n_rows, n_items, keep):
    idx = (np.arange(n_rows, dtype=np.uint64)[:, None] * np.uint64(n_items) + np.arange(n_items, dtype=np.uint64)[None, :])
    return (O.rng_uniform(seed, O.STREAM_VAE_DROPOUT, step, idx).astype(np.float32) < np.float32(keep)).astype(np.float64)


def eps_dense(seed, step, n_rows, Z):
    idx = np.arange(n_rows * Z, dtype=np.uint64).reshape(n_rows, Z)
    return O.rng_normal(seed, O.STREAM_VAE_EPS, step, idx)


def d_masks(seed, step, n, widths, keep):
    out = []
    for stream, w in zip((O.STREAM_D_DROP_A, O.STREAM_D_DROP_B, O.STREAM_D_DROP_C), widths):
        idx = np.arange(n * w, dtype=np.uint64).reshape(n, w)
        out.append((O.rng_uniform(seed, stream, step, idx).astype(np.float32) < np.float32(keep)).astype(np.float64))
    return out


def gen_to_engine(P):
    """oracle dict (TF shapes) -> engine layout list."""
    return [P["Wq0"], P["Wq1"], P["Wp0"], np.ascontiguousarray(P["Wp1"].T), P["bq0"], P["bq1"], P["bp0"], P["bp1"]]


def engine_to_gen(arrs):
    a = [np.asarray(x) for x in arrs]
    return {"Wq0": a[0], "Wq1": a[1], "Wp0": a[2], "Wp1": np.ascontiguousarray(a[3].T), "bq0": a[4], "bq1": a[5],
            "bp0": a[6], "bp1": a[7]}


def disc_to_engine(D):
    return D["emb"], [D[k] for k in O.D_KEYS]


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
