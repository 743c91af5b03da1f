"""Feed conversion of the sess.run facade on the CPU (no kernels are launched): dense X -> CSR rows (+ transposed view),
dense generated_tags mask + id lists -> (row, id) pair lists, and the consistency checks between them."""
import types

import numpy as np
import pytest
import torch

from ltgan.generator import Placeholder
from ltgan.session import Session


def _session(n_items):
    eng = types.SimpleNamespace(I=n_items, device=torch.device("cpu"), new_acts=lambda rows: types.SimpleNamespace(rows=rows))
    return Session(eng)


def test_rows_conversion_matches_dense_matrix():
    I = 50
    rng = np.random.default_rng(0)
    X = (rng.random((7, I)) < 0.2).astype(np.float32)
    X[3] = 0
    s = _session(I)
    batch, B = s._rows(X, with_csc=True)
    indptr, indices = batch.keep[0].numpy(), batch.keep[1].numpy()
    assert B == 7 and batch.n_rows == 7 and batch.keep[2] is None            # all ones -> no values array
    dense = np.zeros_like(X)
    for b in range(B):
        dense[b, indices[indptr[b]:indptr[b + 1]]] = 1
    assert np.array_equal(dense, X)
    slot, uptr, rowidx, csr_pos = (t.numpy() for t in batch.keep[3:7])
    assert batch.c.n_unique == int((X.sum(0) > 0).sum()) == len(uptr) - 1
    for item in np.nonzero(X.sum(0))[0]:
        u = slot[item]
        rows = rowidx[uptr[u]:uptr[u + 1]]
        assert sorted(rows.tolist()) == np.nonzero(X[:, item])[0].tolist()
        assert np.all(indices[csr_pos[uptr[u]:uptr[u + 1]]] == item)
    assert np.all(slot[X.sum(0) == 0] == -1)
    Xv = X * 2.5
    batch2, _ = s._rows(Xv, with_csc=False)
    assert batch2.keep[2] is not None and np.allclose(batch2.keep[2].numpy(), 2.5)
    with pytest.raises(ValueError):
        s._rows(np.zeros((2, I + 1), np.float32), with_csc=False)


def test_fake_pair_list_from_mask_and_ids():
    I = 40
    s = _session(I)
    tags = np.zeros((5, I))
    tags[0, [3, 9]] = 1
    tags[2, [1]] = 1
    tags[4, [7, 8, 30]] = 1
    gen = np.array([3, 9, 1, 7, 8, 30])
    pop = np.array([11, 12, 13, 14, 15, 16])
    ph = {n: Placeholder(n) for n in ("x_generated", "x_popular_g", "generated_tags")}
    feed = {ph["x_generated"]: gen, ph["x_popular_g"]: pop, ph["generated_tags"]: tags}
    names = {p.name: p for p in feed}
    pairs = s._fake_pairs(feed, names, need_rows=True)
    assert pairs.n == 6
    assert pairs.keep[2].numpy().tolist() == [0, 0, 2, 4, 4, 4]              # row of each pair = row-major nonzeros of the mask
    assert pairs.keep[1].numpy().tolist() == gen.tolist() and pairs.keep[0].numpy().tolist() == pop.tolist()
    bad = dict(feed)
    bad[ph["x_generated"]] = np.array([3, 9, 1, 7, 8, 31])
    with pytest.raises(ValueError):
        s._fake_pairs(bad, names, need_rows=True)
    with pytest.raises(KeyError):
        s._fake_pairs({ph["x_generated"]: gen}, {"x_generated": ph["x_generated"]}, need_rows=False)
