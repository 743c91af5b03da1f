"""Synthetic user x item workloads of BASELINE.json: there is no network for real
datasets, and the reference's O(I^2) overlap table cannot be built at these sizes, so the index
path's OUTPUTS are synthesised directly with the same shape statistics as Askubuntu_Sample:

  * item popularity ~ Zipf(s=1) over I items; history length max(5, round(LogNormal(2.2, 0.9)))
    clipped to min(I/2, 2000); items drawn ~ popularity without replacement;
  * popular set = the 10 % most popular items of the generating distribution, niche = the rest; a user's popular / niche
    lists = the history split by that set (users missing either list stay invalid, Q8);
  * real pairs: each niche item of a user paired with one of the user's popular items (uniform:
    timing-only stand-in for the max-overlap partner);
  * candidate set = the user's n niche items + max(2n, 10-n) other niche items drawn ~ popularity;
  * every item id is valid.

Workloads:  ml20m  = 136 000 users x 20 000 items (configs[2]);
            c4     = 200 000 items (configs[3]'s item count), `users` rows generated on demand.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse

from .dataset import IndexData

WORKLOADS = {"ml20m": dict(I=20000, N=136000), "c4": dict(I=200000, N=1000000)}


def _histories(rng, n_users, I):
    pop = 1.0 / np.arange(1, I + 1)
    cdf = np.cumsum(pop / pop.sum())
    L = np.maximum(5, np.rint(rng.lognormal(2.2, 0.9, n_users))).astype(np.int64)
    L = np.minimum(L, min(I // 2, 2000))
    draws = (L * 1.6 + 8).astype(np.int64)
    total = int(draws.sum())
    row = np.repeat(np.arange(n_users), draws)
    item = np.searchsorted(cdf, rng.random(total), side="right").clip(0, I - 1)
    key = row * I + item
    key = np.unique(key)                       # dedup within a user, sorted by (row, item)
    row, item = key // I, key % I
    # truncate each user to L items (keep a random subset so that popular items are not favoured twice)
    order = np.lexsort((rng.random(len(row)), row))
    row, item = row[order], item[order]
    start = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=n_users))])
    rank = np.arange(len(row)) - start[row]
    keep = rank < L[row]
    row, item = row[keep], item[keep]
    order = np.lexsort((item, row))
    return row[order], item[order], pop


def synthetic_index(name, users=None, seed=1234):
    """name: "ml20m", "c4" or "custom:<n_items>"."""
    if name.startswith("custom:"):
        spec = dict(I=int(name.split(":")[1]), N=1000)
    else:
        spec = WORKLOADS[name]
    I = spec["I"]
    N = int(users or spec["N"])
    rng = np.random.default_rng(seed)
    row, item, pop = _histories(rng, N, I)
    train = sparse.csr_matrix((np.ones(len(row), np.float32), (row, item)), shape=(N, I))
    train.sort_indices()
    # popular set = the 10 % of items with the largest generating popularity (Zipf is monotone in the id);
    # sample counts would mark every touched item popular when users << items
    n_pop = max(1, I // 10)
    popular = np.zeros(I, bool)
    popular[:n_pop] = True
    is_pop = popular[item]
    indptr = train.indptr
    npop_u = np.add.reduceat(is_pop.astype(np.int64), indptr[:-1]) if len(item) else np.zeros(N, np.int64)
    npop_u[np.diff(indptr) == 0] = 0
    nnic_u = np.diff(indptr) - npop_u
    ok = (npop_u > 0) & (nnic_u > 0)
    niche_ids = np.nonzero(~popular)[0]
    nic_cdf = np.cumsum(pop[niche_ids] / pop[niche_ids].sum())

    idx = IndexData.__new__(IndexData)
    idx.n_items, idx.N, idx.train, idx.uid0 = I, N, train, 0
    idx.valid_item = np.ones(I, np.uint8)
    idx.user_ok = ok
    pop_ptr, cand_ptr, real_ptr = np.zeros(N + 1, np.int64), np.zeros(N + 1, np.int64), np.zeros(N + 1, np.int64)
    pop_idx, cand_idx, real_nic, real_pop = [], [], [], []
    n_sample = np.where(ok, nnic_u, 0).astype(np.int32)
    for u in range(N):
        if ok[u]:
            it = item[indptr[u]:indptr[u + 1]]
            pm = is_pop[indptr[u]:indptr[u + 1]]
            pl, nl = it[pm], it[~pm]
            n = len(nl)
            want = max(2 * n, 10 - n)
            extra = niche_ids[np.searchsorted(nic_cdf, rng.random(want + 8), side="right").clip(0, len(niche_ids) - 1)]
            extra = np.setdiff1d(np.unique(extra), nl)[:want]
            cand = np.sort(np.concatenate([nl, extra]))
            pop_idx.append(pl)
            cand_idx.append(cand)
            real_nic.append(nl)
            real_pop.append(pl[rng.integers(0, len(pl), n)])
            pop_ptr[u + 1] = pop_ptr[u] + len(pl)
            cand_ptr[u + 1] = cand_ptr[u] + len(cand)
            real_ptr[u + 1] = real_ptr[u] + n
        else:
            pop_ptr[u + 1], cand_ptr[u + 1], real_ptr[u + 1] = pop_ptr[u], cand_ptr[u], real_ptr[u]
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
    idx.pop_ptr, idx.pop_idx = pop_ptr.astype(np.int32), cat(pop_idx)
    idx.cand_ptr, idx.cand_idx = cand_ptr.astype(np.int32), cat(cand_idx)
    idx.real_ptr, idx.real_nic, idx.real_pop = real_ptr.astype(np.int32), cat(real_nic), cat(real_pop)
    idx.n_sample = n_sample
    idx.slot_ptr = np.concatenate([[0], np.cumsum(n_sample)]).astype(np.int32)
    desc = "synthetic %s-shaped: %d users x %d items, %d interactions (Zipf popularity, log-normal history length; SURVEY 8/d2), seed %d" % (
        name, N, I, train.nnz, seed)
    return idx, desc


def write_dataset_dir(out_dir, n_items=20000, n_users=3000, n_eval_users=300, seed=7, niche_frac=0.9, missing_from_item_list=(7, 4242)):
    """A synthetic dataset DIRECTORY in the reference's file formats (train.py:35-38,51,57,77,83-84,91-92; test.py:68-69), so
    that the whole index path (ltgan.data_processing, i.e. the reference's data_processing.py:6-340) can be exercised above
    the SPARSE_OVERLAP_MIN_TAGS switch where the dense I x I overlap table is no longer built:
      item2id.txt "raw<TAB>id", item_list.txt / niche_items.txt / unique_item_id.txt one id per line, item_counts.csv
      "userId,tagId,rating" (the user-tag file of load_overlap_coeff, raw tag ids), train_GAN*.csv / validation_* / test_*
      "uid,sid".  Raw tag id = 100000 + id.  A few ids are left out of item_list.txt (invalid ids, Q9)."""
    import os
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    I = int(n_items)
    row, item, _ = _histories(rng, n_users + 2 * n_eval_users, I)
    n_pop = max(1, int(round(I * (1.0 - niche_frac))))
    is_pop_item = np.zeros(I, bool)
    is_pop_item[:n_pop] = True
    raw = lambda ids: np.asarray(ids, np.int64) + 100000

    def csv(name, u, s):
        with open(os.path.join(out_dir, name), "w") as f:
            f.write("uid,sid\n")
            np.savetxt(f, np.stack([np.asarray(u, np.int64), np.asarray(s, np.int64)], 1), fmt="%d", delimiter=",")

    tr = row < n_users
    csv("train_GAN.csv", row[tr], item[tr])
    csv("train_GAN_popular.csv", row[tr & is_pop_item[item]], item[tr & is_pop_item[item]])
    csv("train_GAN_niche.csv", row[tr & ~is_pop_item[item]], item[tr & ~is_pop_item[item]])
    for name, lo in (("validation", n_users), ("test", n_users + n_eval_users)):
        sel = (row >= lo) & (row < lo + n_eval_users)
        held = sel & (rng.random(len(row)) < 0.2)
        csv(name + "_tr.csv", row[sel & ~held], item[sel & ~held])
        csv(name + "_te.csv", row[held], item[held])
    # the user-tag file: every (user, tag) interaction of the training users, plus "background" users (ids beyond the training
    # range, 5 tags each) that cover the tags no training user touched -- the reference indexes OVERLAP_COEFFS[tag] for every
    # niche tag (data_processing.py:200-211) and raises KeyError for a tag without users
    cu, ct = row[tr], item[tr]
    untouched = np.setdiff1d(np.arange(I), np.unique(ct))
    bu = n_users + 2 * n_eval_users + np.arange(len(untouched)) // 5
    cu, ct = np.concatenate([cu, bu]), np.concatenate([ct, untouched])
    with open(os.path.join(out_dir, "item_counts.csv"), "w") as f:
        f.write("userId,tagId,rating\n")
        np.savetxt(f, np.stack([cu, raw(ct), np.ones(len(cu), np.int64)], 1), fmt="%d", delimiter=",")
    with open(os.path.join(out_dir, "item2id.txt"), "w") as f:
        for i in range(I):
            f.write("%d\t%d\n" % (100000 + i, i))
    skip = set(int(x) for x in missing_from_item_list if x < I)
    with open(os.path.join(out_dir, "item_list.txt"), "w") as f:
        for i in range(I):
            if i not in skip:
                f.write("%d\n" % (100000 + i))
    with open(os.path.join(out_dir, "niche_items.txt"), "w") as f:
        for i in range(n_pop, I):
            f.write("%d\n" % (100000 + i))
    with open(os.path.join(out_dir, "unique_item_id.txt"), "w") as f:
        for i in range(I):
            f.write("%d\n" % i)
    return out_dir
