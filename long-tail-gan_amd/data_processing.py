"""Index path: the loaders of the reference's Codes/data_processing.py, same names and return
values, restated with array arithmetic (bit-exact integer / float64 outputs; pinned against the
reference's own module by tests/test_index_path.py and tests/golden/askubuntu_golden.npz).

These run once on the host before training (train.py:56-113); their outputs are uploaded to HBM
by ltgan.dataset.  The overlap table is a dense I x I float64 matrix up to SPARSE_OVERLAP_MIN_TAGS tags (the
reference builds a dict of dicts); above that `load_overlap_coeff` returns the co-occurrence CSR form
(`SparseOverlap`) and `load_vectors` / `load_items_to_sample` materialise only per-user sub-blocks
with the dense path's arithmetic -- bit-exact against the dense form (tests/test_index_path.py).
"""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import sparse


def load_train_data(csv_file, n_items):
    """data_processing.py:6-17 -> (csr float32 [max(uid)+1, n_items], min uid)."""
    tp = pd.read_csv(csv_file)
    uid = tp["uid"].to_numpy()
    sid = tp["sid"].to_numpy()
    n_users = int(uid.max()) + 1
    data = sparse.csr_matrix((np.ones(len(uid), dtype=uid.dtype), (uid, sid)), dtype="float32", shape=(n_users, n_items))
    return data, tp["uid"].min()


def load_tr_te_data(csv_file_tr, csv_file_te, n_items):
    """data_processing.py:20-37 -> (csr float64 tr, csr float64 te, start uid)."""
    tr = pd.read_csv(csv_file_tr)
    te = pd.read_csv(csv_file_te)
    lo = min(tr["uid"].min(), te["uid"].min())
    hi = max(tr["uid"].max(), te["uid"].max())
    shape = (hi - lo + 1, n_items)

    def mk(tp):
        r = tp["uid"].to_numpy() - lo
        return sparse.csr_matrix((np.ones(len(r), dtype=r.dtype), (r, tp["sid"].to_numpy())), dtype="float64", shape=shape)

    return mk(tr), mk(te), lo


def _read_show2id(path):
    show2id = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split("\t")
            show2id[parts[0]] = parts[1]
    return show2id


def load_pop_niche_tags(show2id_path, item_list_path, niche_tags_path, n_items):
    """data_processing.py:275-340 -> (SHOW2ID, IDs_present, NICHE_TAGS, ALL_TAGS, OTHER_TAGS)."""
    show2id = _read_show2id(show2id_path)
    present = set()
    with open(item_list_path, "r", encoding="utf-8") as f:
        for line in f:
            key = line.strip()
            if key in show2id:
                present.add(show2id[key])
    niche = set()
    with open(niche_tags_path, "r", encoding="utf-8") as f:
        for line in f:
            key = line.strip()
            if key in show2id and show2id[key] in present:
                niche.add(int(show2id[key]))
    all_tags = list(range(n_items))
    other = np.asarray(sorted(set(all_tags) - niche))
    return show2id, present, niche, all_tags, other


def load_item_one_hot_features(item_list_path, SHOW2ID, n_items):
    """data_processing.py:40-70 -> (dict id -> one-hot list, FEATURE_LEN, array).  Only the dict's key
    set (the validity filter, Q9) and FEATURE_LEN are live; the array is dead in the reference (Q7) and
    is returned with the same (mis-aligned) row selection for completeness."""
    feat = {}
    flen = 0
    with open(item_list_path, "r", encoding="utf-8") as f:
        for line in f:
            key = line.strip()
            if key not in SHOW2ID:
                continue
            idx = int(SHOW2ID[key])
            row = [0] * n_items
            row[idx] = 1
            feat[idx] = row
            flen = n_items
    arr = np.array([feat[i] for i in range(len(feat)) if i in feat])
    return feat, flen, arr


def load_valid_item_ids(item_list_path, SHOW2ID):
    """The key set of ITEM_FEATURE_DICT (data_processing.py:54-59) without its O(I^2) one-hot rows: the ids of the
    item_list.txt lines that appear in SHOW2ID.  It is the only part of load_item_one_hot_features the training path
    consumes (validity filter of train.py:240-243 and data_processing.py:258-260, Q9)."""
    ids = set()
    with open(item_list_path, "r", encoding="utf-8") as f:
        for line in f:
            key = line.strip()
            if key in SHOW2ID:
                ids.add(int(SHOW2ID[key]))
    return ids


def load_user_items(csv_file_path):
    """data_processing.py:72-96 -> dict uid -> list of sids in FILE order."""
    tp = pd.read_csv(csv_file_path)
    uid = tp.iloc[:, 0].to_numpy()
    sid = tp.iloc[:, 1].to_numpy()
    out = {}
    for u, s in zip(uid.tolist(), sid.tolist()):
        out.setdefault(u, []).append(s)
    return out


def _tag_membership(show2id_path, user_tag_matrix_path):
    """(M, size): binary user x tag CSR of the user-tag file restricted to ids of item2id.txt (TAG_SETS of
    data_processing.py:110-140 are sets: a (user, tag) pair counts once), |U_a| per tag."""
    show2id = _read_show2id(show2id_path)
    tp = pd.read_csv(user_tag_matrix_path)
    users = tp.iloc[:, 0].astype(str).to_numpy()
    tags = tp.iloc[:, 1].astype(str).to_numpy()
    keep = np.array([t in show2id for t in tags])
    users, tags = users[keep], tags[keep]
    tid = np.array([int(show2id[t]) for t in tags], dtype=np.int64)
    _, uidx = np.unique(users, return_inverse=True)
    n_tags = int(max(int(v) for v in show2id.values())) + 1
    M = sparse.csr_matrix((np.ones(len(tid), np.int64), (uidx, tid)), shape=(uidx.max() + 1, n_tags))
    M.sum_duplicates()
    M.data[:] = 1
    size = np.asarray(M.sum(axis=0)).reshape(-1).astype(np.int64)
    return M, size


def overlap_matrix(show2id_path, user_tag_matrix_path):
    """Dense float64 matrix OC[a, b] = |U_a & U_b| / min(|U_a|, |U_b|) (data_processing.py:100-164),
    NaN for ids that never occur in the user-tag file."""
    M, size = _tag_membership(show2id_path, user_tag_matrix_path)
    inter = np.asarray((M.T @ M).todense(), dtype=np.int64)
    denom = np.minimum(size[:, None], size[None, :]).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        oc = (inter * 1.0) / (1.0 * denom)
    missing = size == 0
    oc[missing, :] = np.nan
    oc[:, missing] = np.nan
    return oc


class SparseOverlap:
    """The same coefficients without the O(I^2) table: the co-occurrence counts |U_a & U_b| as a CSR
    matrix (only pairs of tags that share a user are stored) + |U_a|.  `block(rows, cols)` materialises the small
    sub-blocks the two consumers need with exactly the dense path's arithmetic (int64 count * 1.0 / float64 min)."""

    def __init__(self, show2id_path, user_tag_matrix_path):
        M, self.size = _tag_membership(show2id_path, user_tag_matrix_path)
        self.inter = (M.T @ M).tocsr()
        self.inter.sort_indices()
        self.present = self.size > 0
        self.n = len(self.size)

    def block(self, rows, cols):
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        cnt = np.asarray(self.inter[rows][:, cols].todense(), dtype=np.int64)
        denom = np.minimum(self.size[rows][:, None], self.size[cols][None, :]).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            oc = (cnt * 1.0) / (1.0 * denom)
        oc[~self.present[rows], :] = np.nan
        oc[:, ~self.present[cols]] = np.nan
        return oc

    def row_max(self, rows):
        """max over `rows` of OC[row, :] as a dense length-n vector (0 where no row shares a user with the column),
        and whether any row is an id without users (its coefficients are undefined: KeyError in the reference)."""
        rows = np.asarray(rows, dtype=np.int64)
        out = np.zeros(self.n, dtype=np.float64)
        ip, ix, dt = self.inter.indptr, self.inter.indices, self.inter.data
        for r in rows.tolist():
            c = ix[ip[r]:ip[r + 1]]
            v = (dt[ip[r]:ip[r + 1]].astype(np.int64) * 1.0) / (1.0 * np.minimum(self.size[r], self.size[c]).astype(np.float64))
            np.maximum.at(out, c, v)
        return out, bool((~self.present[rows]).any())


class _OverlapView(dict):
    """dict-of-dicts facade over the dense matrix: OVERLAP_COEFFS[a][b] like the reference."""

    def __init__(self, oc):
        super().__init__()
        self.matrix = oc
        present = ~np.isnan(np.diag(oc))
        for a in np.nonzero(present)[0].tolist():
            self[a] = _Row(oc[a], present)


class _Row:
    def __init__(self, row, present):
        self.row, self.present = row, present

    def __getitem__(self, b):
        if not self.present[b]:
            raise KeyError(b)
        return float(self.row[b])

    def __len__(self):
        return int(self.present.sum())


class _SparseOverlapView:
    """OVERLAP_COEFFS[a][b] over SparseOverlap (rows are produced on demand)."""

    def __init__(self, sp):
        self.sparse = sp

    def __contains__(self, a):
        return 0 <= a < self.sparse.n and bool(self.sparse.present[a])

    def __getitem__(self, a):
        if a not in self:
            raise KeyError(a)
        sp = self.sparse
        return _Row(sp.block([a], np.arange(sp.n))[0], sp.present)

    def __len__(self):
        return int(self.sparse.present.sum())


# above this many tags the dense I x I float64 table is replaced by the sparse co-occurrence form
SPARSE_OVERLAP_MIN_TAGS = 8192


def load_overlap_coeff(show2id_path, user_tag_matrix_path, sparse_form=None):
    """data_processing.py:110-167 -> OVERLAP_COEFFS[a][b] (float).  sparse_form: None = by size, True / False = force."""
    if sparse_form is None:
        sparse_form = int(max(int(v) for v in _read_show2id(show2id_path).values())) + 1 > SPARSE_OVERLAP_MIN_TAGS
    if sparse_form:
        return _SparseOverlapView(SparseOverlap(show2id_path, user_tag_matrix_path))
    return _OverlapView(overlap_matrix(show2id_path, user_tag_matrix_path))


def _matrix(OVERLAP_COEFFS):
    if isinstance(OVERLAP_COEFFS, _OverlapView):
        return OVERLAP_COEFFS.matrix
    n = max(OVERLAP_COEFFS) + 1
    oc = np.full((n, n), np.nan)
    for a, row in OVERLAP_COEFFS.items():
        for b, v in row.items():
            oc[a, b] = v
    return oc


def load_vectors(user_popular_data, user_niche_data, OVERLAP_COEFFS, ITEM_FEATURE_DICT, N):
    """data_processing.py:227-271: for each niche item of a user (file order) the popular item of the
    user with the largest overlap (first maximum wins: strict '>' at :254); pairs touching an id
    outside ITEM_FEATURE_DICT are dropped (:258-260)."""
    sp = OVERLAP_COEFFS.sparse if isinstance(OVERLAP_COEFFS, _SparseOverlapView) else None
    oc = None if sp is not None else _matrix(OVERLAP_COEFFS)
    n_ids = sp.n if sp is not None else oc.shape[0]
    valid = np.zeros(n_ids, bool)
    valid[[k for k in ITEM_FEATURE_DICT if 0 <= k < n_ids]] = True      # only the KEYS are used (a dict or a set of ids)
    if sp is not None:
        return _load_vectors_sparse(user_popular_data, user_niche_data, sp, valid, N)
    x_niche, x_pop = {}, {}
    for u in range(N):
        if u not in user_popular_data or u not in user_niche_data:
            continue
        pops = np.asarray(user_popular_data[u])
        nics = np.asarray(user_niche_data[u])
        sub = sp.block(nics, pops) if sp is not None else oc[np.ix_(nics, pops)]
        if np.isnan(sub).any():
            raise KeyError("overlap coefficient missing for user %d" % u)  # the reference raises KeyError
        best = pops[np.argmax(sub, axis=1)]  # argmax returns the FIRST maximum
        ok = valid[nics] & valid[best]
        x_niche[u] = nics[ok].tolist()
        x_pop[u] = best[ok].tolist()
    return x_niche, x_pop


def _load_vectors_sparse(user_popular_data, user_niche_data, sp, valid, N):
    """load_vectors over the co-occurrence CSR, all users at once: every (niche item, popular item) pair of every user is
    scored in one vectorised pass (int64 count * 1.0 / float64 min -- the dense path's arithmetic), then a segmented
    FIRST-maximum per (user, niche item) (strict '>' at data_processing.py:254)."""
    users = [u for u in range(N) if u in user_popular_data and u in user_niche_data]
    x_niche, x_pop = {}, {}
    if not users:
        return x_niche, x_pop
    nic_l = [np.asarray(user_niche_data[u], dtype=np.int64) for u in users]
    pop_l = [np.asarray(user_popular_data[u], dtype=np.int64) for u in users]
    n_u = np.array([len(a) for a in nic_l])
    p_u = np.array([len(a) for a in pop_l])
    # one segment per (user, niche item) holding that user's popular items in file order
    seg_len = np.repeat(p_u, n_u)
    seg_nic = np.concatenate(nic_l)
    seg_start = np.concatenate([[0], np.cumsum(seg_len)])
    pop_off = np.concatenate([[0], np.cumsum(p_u)])
    pop_cat = np.concatenate(pop_l)
    seg_user = np.repeat(np.arange(len(users)), n_u)
    pos_in_seg = np.arange(seg_start[-1]) - np.repeat(seg_start[:-1], seg_len)
    pair_pop = pop_cat[np.repeat(pop_off[seg_user], seg_len) + pos_in_seg]
    pair_nic = np.repeat(seg_nic, seg_len)
    if not (sp.present[pair_pop].all() and sp.present[pair_nic].all()):
        raise KeyError("overlap coefficient missing for a user's popular / niche item")   # the reference raises KeyError
    cnt = np.asarray(sp.inter[pair_nic, pair_pop], dtype=np.int64).reshape(-1)
    score = (cnt * 1.0) / (1.0 * np.minimum(sp.size[pair_nic], sp.size[pair_pop]).astype(np.float64))
    nz = seg_len > 0
    best = np.full(len(seg_len), -1, np.int64)
    if nz.any():
        mx = np.maximum.reduceat(score, seg_start[:-1][nz])
        is_max = score == np.repeat(mx, seg_len[nz])
        first = np.minimum.reduceat(np.where(is_max, pos_in_seg, np.iinfo(np.int64).max), seg_start[:-1][nz])
        best[nz] = pair_pop[seg_start[:-1][nz] + first]
    ok = valid[seg_nic] & valid[np.maximum(best, 0)] & (best >= 0)
    u_off = np.concatenate([[0], np.cumsum(n_u)])
    for k, u in enumerate(users):
        sl = slice(u_off[k], u_off[k + 1])
        o = ok[sl]
        x_niche[u] = seg_nic[sl][o].tolist()
        x_pop[u] = best[sl][o].tolist()
    return x_niche, x_pop


def _candidates_sorted_order(sp, nics, cur, niche_sorted, want):
    """load_items_to_sample for one user when `NICHE_TAGS - curr_niche_tags` iterates in ascending id order: the stable
    descending sort of data_processing.py:214 = tags with a positive score by (score desc, id asc), then zero-score tags by
    id asc.  Only the tags that share a user with one of the user's niche items can have a positive score."""
    rows = np.asarray(nics, dtype=np.int64)
    if (~sp.present[rows]).any():
        raise KeyError("overlap coefficient missing for a niche item of the user")
    ip, ix, dt = sp.inter.indptr, sp.inter.indices, sp.inter.data
    cols, vals = [], []
    for r in rows.tolist():
        c = ix[ip[r]:ip[r + 1]]
        cols.append(c)
        vals.append((dt[ip[r]:ip[r + 1]].astype(np.int64) * 1.0) / (1.0 * np.minimum(sp.size[r], sp.size[c]).astype(np.float64)))
    cols = np.concatenate(cols).astype(np.int64)
    vals = np.concatenate(vals)
    cur_arr = np.asarray(sorted(cur), dtype=np.int64)
    n_others = len(niche_sorted) - int(np.isin(cur_arr, niche_sorted, assume_unique=True).sum())
    k = min(want, n_others)
    # max score per co-occurring tag, restricted to niche tags outside the user's own
    keep = np.isin(cols, niche_sorted) & ~np.isin(cols, cur_arr)
    cols, vals = cols[keep], vals[keep]
    order = np.lexsort((-vals, cols))
    cols, vals = cols[order], vals[order]
    first = np.concatenate([[True], cols[1:] != cols[:-1]]) if len(cols) else np.zeros(0, bool)
    pid, pscore = cols[first], vals[first]                        # ascending id, max score each
    pos = pscore > 0
    pid, pscore = pid[pos], pscore[pos]
    sel = pid[np.argsort(-pscore, kind="stable")][:k]             # stable: equal scores keep ascending id
    if len(sel) < k:
        need = k - len(sel)
        excl = np.union1d(cur_arr, pid)
        head = niche_sorted[:need + len(excl)]                    # enough of the smallest ids to survive the exclusion
        zeros = head[~np.isin(head, excl, assume_unique=True)][:need]
        sel = np.concatenate([sel, zeros])
    picked = sorted([int(t) for t in nics] + sel.tolist())
    return np.asarray(picked)


def _cpython_set_difference_is_sorted(so_sorted, n_other):
    """True when CPython's `so - other` (so: a set of non-negative ints, other: a set of n_other ints) is GUARANTEED to iterate
    in ascending order: hash(i) == i, so an int sits in slot i & mask of the table, and when every element is below the table
    size there are no collisions and table order is value order.  Table size of the result (Objects/setobject.c, 3.8+):
      * len(so) >> 2 > len(other): the result is a COPY of so (set_merge into an empty table resized for 2 * used entries:
        the smallest power of two above 2 * len(so)) from which other's elements are discarded;
      * otherwise the elements are inserted one by one (growth by 4x, 2x above 50 000 entries, whenever the table is 3/5 full)."""
    n = len(so_sorted)
    if n == 0:
        return True
    top = int(so_sorted[-1])
    if (n >> 2) > n_other:
        size = 8
        while size <= 2 * n:
            size <<= 1
        return top < size
    mask, fill = 7, 0
    for _ in range(max(0, n - n_other)):       # (the loop runs at most len(so) times, once per distinct other-count)
        fill += 1
        if fill * 5 >= mask * 3:
            minused = fill * 2 if fill > 50000 else fill * 4
            size = 8
            while size <= minused:
                size <<= 1
            mask = size - 1
    return top <= mask


def load_items_to_sample(user_popular_data, user_niche_data, NICHE_TAGS, OVERLAP_COEFFS, N):
    """data_processing.py:170-224: candidate set = the user's niche items + the top
    max(2n, 10-n) other niche items ranked by their max overlap with any of the user's niche items;
    ties keep the iteration order of the Python set difference (stable sort, :214) -- reproduced by
    performing that very set operation; ascending ids."""
    sp = OVERLAP_COEFFS.sparse if isinstance(OVERLAP_COEFFS, _SparseOverlapView) else None
    oc = None if sp is not None else _matrix(OVERLAP_COEFFS)
    out = {}
    niche_set = set(NICHE_TAGS)
    niche_sorted = np.asarray(sorted(niche_set), dtype=np.int64)
    sorted_ok = {}          # number of distinct niche items of a user -> does `niche_set - cur` iterate in ascending order?
    all_present = sp is not None and len(niche_sorted) > 0 and int(niche_sorted[-1]) < sp.n and bool(sp.present[niche_sorted].all())
    for u in range(N):
        if u not in user_popular_data or u not in user_niche_data:
            continue
        nics = user_niche_data[u]
        n = len(nics)
        want = max(2 * n, 10 - n)
        cur = set()
        for t in nics:
            cur.add(t)
        if sp is not None and all_present:
            # large tag sets: when the set difference provably iterates in ascending id order, the candidates follow from the
            # co-occurring tags alone (score > 0: by score, ties by id) and the smallest remaining ids (score 0) -- no
            # O(|niche|) set operation per user
            if len(cur) not in sorted_ok:
                sorted_ok[len(cur)] = _cpython_set_difference_is_sorted(niche_sorted, len(cur))
            if sorted_ok[len(cur)]:
                out[u] = _candidates_sorted_order(sp, nics, cur, niche_sorted, want)
                continue
        others = np.asarray(list(niche_set - cur), dtype=np.int64)  # CPython set-difference order
        picked = [int(t) for t in nics]
        if len(others) and sp is not None:
            # sparse form: only tags that share a user with one of the user's niche items score > 0; the stable
            # descending sort of the reference (:214) = positives by (score desc, set order), then zeros in set order
            full, bad = sp.row_max(nics)
            if bad or not sp.present[others].all():
                raise KeyError("overlap coefficient missing for user %d" % u)
            score = full[others]
            k = min(want, len(others))
            pos = np.nonzero(score > 0)[0]
            order = pos[np.argsort(-score[pos], kind="stable")][:k]
            if len(order) < k:
                order = np.concatenate([order, np.nonzero(score <= 0)[0][:k - len(order)]])
            picked += others[order].tolist()
        elif len(others):
            score = oc[np.ix_(np.asarray(nics), others)].max(axis=0)
            if np.isnan(score).any():
                raise KeyError("overlap coefficient missing for user %d" % u)
            order = np.argsort(-score, kind="stable")[:min(want, len(others))]
            picked += others[order].tolist()
        picked.sort()
        out[u] = np.asarray(picked)
    return out
