"""Device-resident training data: runs the index path (ltgan.data_processing == the reference's
train.py:56-113 set-up) once on the host and lays its outputs out in HBM as flat CSR-style arrays,
one contiguous slice per batch of BATCH_SIZE users (train.py:192-198: batches are consecutive
user-id ranges, fixed for the whole run).

Per batch b (users [b*BS, min((b+1)*BS, N))):
  X rows          indptr/indices               (replaces the dense float32 [B,I] feed)
  X^T view        slot[b] / uptr / uitem / rowidx / csr_pos (entries grouped by item: sparse gradient of W_q0)
  real pairs      x_popular_n / x_niche        (train.py:223-224; static)
  sampler inputs  candidates, popular lists, to_sample, output slots (train.py:213-227)
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _cabi as cabi
from . import data_processing as dp
from .engine import CsrRows, Pairs, _ptr

DATASET_FILES = ["item2id.txt", "niche_items.txt", "item_counts.csv", "item_list.txt", "unique_item_id.txt",
                 "train_GAN.csv", "validation_tr.csv", "validation_te.csv", "train_GAN_popular.csv", "train_GAN_niche.csv",
                 "test_tr.csv", "test_te.csv"]   # train.py:35-38,51,57,77,83-84,91-92; test.py:68-69


def materialize_askubuntu(raw_npz, out_dir):
    """Re-create the Askubuntu_Sample dataset directory from tests/golden/askubuntu_raw.npz (the
    dataset's files stored as integer arrays)."""
    r = np.load(raw_npz)
    os.makedirs(out_dir, exist_ok=True)

    def pairs(name):
        with open(os.path.join(out_dir, name + ".csv"), "w") as f:
            f.write("uid,sid\n")
            np.savetxt(f, np.stack([r[name + "_uid"].astype(np.int64), r[name + "_sid"].astype(np.int64)], 1), fmt="%d", delimiter=",")

    for name in ["train_GAN", "train_GAN_popular", "train_GAN_niche", "validation_tr", "validation_te", "test_tr", "test_te"]:
        pairs(name)
    with open(os.path.join(out_dir, "item_counts.csv"), "w") as f:
        f.write("userId,tagId,rating\n")
        for u, t, x in zip(r["item_counts_userId"].tolist(), r["item_counts_tagId"].tolist(), r["item_counts_rating"].tolist()):
            f.write("%d,%d,%s\n" % (u, t, repr(float(x))))
    with open(os.path.join(out_dir, "item2id.txt"), "w") as f:
        for a, b in zip(r["item2id_raw"].tolist(), r["item2id_id"].tolist()):
            f.write("%d\t%d\n" % (a, b))
    for name in ["item_list", "niche_items", "unique_item_id"]:
        with open(os.path.join(out_dir, name + ".txt"), "w") as f:
            for a in r[name].tolist():
                f.write("%d\n" % a)
    return out_dir


def count_items(pro_dir):
    """generator.py:6-11 / train.py:56-61: n_items = number of lines of unique_item_id.txt."""
    with open(os.path.join(pro_dir, "unique_item_id.txt"), "r") as f:
        return sum(1 for _ in f)


class IndexData:
    """Host-side result of the index path in flat-array form (what gets uploaded)."""

    def __init__(self, n_items, train_csr, uid_start_idx, user_pop, user_niche, x_niche, x_pop_n, cand, valid_ids):
        self.n_items = n_items
        self.N = train_csr.shape[0]
        tr = train_csr.tocsr()
        tr.sort_indices()
        self.train = tr
        self.uid0 = int(uid_start_idx)
        N = self.N
        self.valid_item = np.zeros(n_items, np.uint8)
        self.valid_item[np.asarray(sorted(valid_ids), np.int64)] = 1
        # user u (row index) <-> key u + uid_start_idx in the dicts (train.py:213)
        self.user_ok = np.zeros(N, bool)
        pop_ptr, pop_idx, cand_ptr, cand_idx, real_ptr, real_nic, real_pop = [0], [], [0], [], [0], [], []
        n_sample = np.zeros(N, np.int32)
        for r in range(N):
            u = r + self.uid0
            ok = u in user_pop and u in user_niche
            self.user_ok[r] = ok
            if ok:
                pop_idx += list(user_pop[u])
                cand_idx += list(cand[u])
                real_nic += list(x_niche[u])
                real_pop += list(x_pop_n[u])
                n_sample[r] = len(user_niche[u])      # to_sample = len(curr_niche_vectors), train.py:227
            pop_ptr.append(len(pop_idx))
            cand_ptr.append(len(cand_idx))
            real_ptr.append(len(real_nic))
        i32 = lambda a: np.asarray(a, np.int32)
        self.pop_ptr, self.pop_idx = i32(pop_ptr), i32(pop_idx)
        self.cand_ptr, self.cand_idx = i32(cand_ptr), i32(cand_idx)
        self.real_ptr, self.real_nic, self.real_pop = i32(real_ptr), i32(real_nic), i32(real_pop)
        self.n_sample = n_sample
        self.slot_ptr = np.concatenate([[0], np.cumsum(n_sample)]).astype(np.int32)

    @classmethod
    def from_dir(cls, dataset_dir, verbose=False):
        """The set-up section of train_GAN (train.py:33-113) with the build's loaders."""
        d = dataset_dir
        say = (lambda *a: print(*a, flush=True)) if verbose else (lambda *a: None)
        n_items = count_items(d)
        say("Loading Items...")
        show2id, ids_present, niche, _all, _other = dp.load_pop_niche_tags(os.path.join(d, "item2id.txt"), os.path.join(d, "item_list.txt"),
                                                                       os.path.join(d, "niche_items.txt"), n_items)
        # only the key set of ITEM_FEATURE_DICT is live (Q7/Q9): no n_items x n_items one-hot table on the host
        fdict = dp.load_valid_item_ids(os.path.join(d, "item_list.txt"), show2id)
        say("Loading Training Interaction Matrix...")
        train, uid0 = dp.load_train_data(os.path.join(d, "train_GAN.csv"), n_items)
        upop = dp.load_user_items(os.path.join(d, "train_GAN_popular.csv"))
        unic = dp.load_user_items(os.path.join(d, "train_GAN_niche.csv"))
        say("Loading item overlap coefficients....")
        oc = dp.load_overlap_coeff(os.path.join(d, "item2id.txt"), os.path.join(d, "item_counts.csv"))
        N = train.shape[0]
        xn, xp = dp.load_vectors(upop, unic, oc, fdict, N)
        say("Loading Items to Sample....")
        cand = dp.load_items_to_sample(upop, unic, niche, oc, N)
        return cls(n_items, train, uid0, upop, unic, xn, xp, cand, sorted(fdict))


def batch_csc(tr, lo, hi, n_items):
    """Transposed view of rows [lo, hi) of a CSR matrix: entries grouped by item.
    Returns (slot [I] item -> group or -1, uptr [nu+1], local row per entry, csr position per entry)."""
    beg, end = tr.indptr[lo], tr.indptr[hi]
    idx = tr.indices[beg:end].astype(np.int64)
    row_of = np.repeat(np.arange(hi - lo), np.diff(tr.indptr[lo:hi + 1]))
    pos = np.arange(beg, end, dtype=np.int64)
    order = np.lexsort((row_of, idx))
    uitem, counts = np.unique(idx, return_counts=True)
    slot = np.full(n_items, -1, np.int32)
    slot[uitem] = np.arange(len(uitem), dtype=np.int32)
    uptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return slot, uptr, row_of[order].astype(np.int32), pos[order].astype(np.int32)


SLOT_CACHE_BYTES = 1 << 30     # budget of the per-batch item -> gradient-row maps (n_batches x I int32)


class DeviceData:
    """IndexData uploaded to HBM + per-batch views (ctypes structs with the right offsets)."""

    def __init__(self, idx: IndexData, batch_size, device, item_lo=0, item_hi=None, slot_cache=None):
        """item_lo/item_hi: this rank's item slab (multi-GPU item sharding).  Only the interaction matrix is
        cut (columns [item_lo, item_hi), re-indexed from 0); pair / candidate / popular lists keep global ids
        and are identical on every rank.
        slot_cache: keep the per-batch item -> gradient-row maps (n_batches x I int32) on the device; None = for item slabs
        below 65 536 items whose maps take at most SLOT_CACHE_BYTES in total (two launches = ~16 us fewer per G step of the
        dense W_q0 sweep; the maps are first built on the host, so the bound is on host AND device memory).  Without
        it the library rebuilds the map of a batch in its workspace each step -- measured FASTER at I = 200 000 (50.3 vs
        56.5 ms per 32-step epoch: the one map in the workspace stays in L2, 64 cold 800-KB maps do not) and it removes
        the n_batches x I memory (8 GB at 10 000 batches)."""
        self.idx, self.BS, self.device = idx, int(batch_size), torch.device(device)
        N = idx.N
        self.item_lo, self.item_hi = int(item_lo), int(idx.n_items if item_hi is None else item_hi)
        I = self.item_hi - self.item_lo
        self.N, self.I, self.I_global = N, I, idx.n_items
        self.n_batches = (N + self.BS - 1) // self.BS
        up = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a if dt is None else a.astype(dt))).to(self.device)
        full = idx.train
        self.row_norm2 = up(np.asarray(full.multiply(full).sum(axis=1)).reshape(-1), np.float32)   # sum x^2 over the FULL row
        if (self.item_lo, self.item_hi) != (0, idx.n_items):
            tr = full[:, self.item_lo:self.item_hi].tocsr()
            tr.sort_indices()
        else:
            tr = full
        self.local_csr = tr
        self.indptr = up(tr.indptr, np.int32)
        self.indices = up(tr.indices, np.int32)
        ones = np.all(tr.data == 1.0)
        self.values = None if ones else up(tr.data, np.float32)
        slots, uptrs, uitems, rowidx, cpos, ent_off, uptr_off = [], [], [], [], [], [0], [0]
        if slot_cache is None:
            slot_cache = I < 65536 and self.n_batches * I * 4 <= SLOT_CACHE_BYTES
        for b in range(self.n_batches):
            lo, hi = b * self.BS, min(N, (b + 1) * self.BS)
            sl, up_, ri, ps = batch_csc(tr, lo, hi, I)
            if slot_cache:
                slots.append(sl)
            uptrs.append(up_)
            uitems.append(np.append(np.flatnonzero(sl >= 0), -1).astype(np.int32))   # (padded to uptr's length: one offset serves both)
            rowidx.append(ri)
            cpos.append(ps)
            ent_off.append(ent_off[-1] + len(ri))
            uptr_off.append(uptr_off[-1] + len(up_))
        self.slot = up(np.concatenate(slots)) if slot_cache else None
        self.uptr = up(np.concatenate(uptrs))
        self.uitem = None if os.environ.get("LTGAN_NO_UITEM") else up(np.concatenate(uitems))   # (measurement switch: the longer index chain)
        self.rowidx = up(np.concatenate(rowidx) if ent_off[-1] else np.zeros(1, np.int32))
        self.csr_pos = up(np.concatenate(cpos) if ent_off[-1] else np.zeros(1, np.int32))
        self.ent_off, self.uptr_off = ent_off, uptr_off
        self.pop_ptr, self.pop_idx = up(idx.pop_ptr), up(idx.pop_idx if len(idx.pop_idx) else np.zeros(1, np.int32))
        self.cand_ptr, self.cand_idx = up(idx.cand_ptr), up(idx.cand_idx if len(idx.cand_idx) else np.zeros(1, np.int32))
        self.n_sample, self.slot_ptr = up(idx.n_sample), up(idx.slot_ptr)
        self.valid_item = up(idx.valid_item)
        self.real_nic = up(idx.real_nic if len(idx.real_nic) else np.zeros(1, np.int32))
        self.real_pop = up(idx.real_pop if len(idx.real_pop) else np.zeros(1, np.int32))
        n_slots = int(idx.slot_ptr[-1])
        self.n_slots = n_slots
        # fake pairs of the current global epoch (train.py:192-269 caches them for all sub-epochs, Q6)
        self.fake_gen = torch.full((max(1, n_slots),), -1, dtype=torch.int32, device=self.device)
        self.fake_pop = torch.full((max(1, n_slots),), -1, dtype=torch.int32, device=self.device)
        slot_row = np.repeat(np.arange(N) % self.BS, idx.n_sample).astype(np.int32)   # local row of each slot
        self.fake_row = up(slot_row if n_slots else np.zeros(1, np.int32))
        self.fake_cnt = torch.zeros(self.n_batches, dtype=torch.int32, device=self.device)
        cand_len = np.diff(idx.cand_ptr)
        self.max_rows = min(self.BS, N)
        self._views = [self._make_view(b, cand_len) for b in range(self.n_batches)]
        self._spans = {}
        self._towers = None
        self.max_pairs = max(v["n_real"] + v["n_slots"] for v in self._views) if self._views else 0

    def _make_view(self, b, cand_len):
        idx = self.idx
        lo, hi = b * self.BS, min(self.N, (b + 1) * self.BS)
        I = self.I
        batch = CsrRows(self.indptr, self.indices, lo, hi, values=self.values, slot=self.slot, uptr=self.uptr, rowidx=self.rowidx,
                        csr_pos=self.csr_pos, n_unique=self.uptr_off[b + 1] - self.uptr_off[b] - 1, slot_off=b * I,
                        uptr_off=self.uptr_off[b], ent_off=self.ent_off[b], row_norm2=self.row_norm2, uitem=self.uitem)
        # csr_pos holds ABSOLUTE positions and rowidx LOCAL rows: both are already relative to the arrays given
        r0, r1 = int(idx.real_ptr[lo]), int(idx.real_ptr[hi])
        real = Pairs(self.real_pop, self.real_nic, None, n=r1 - r0, off=r0)
        s0, s1 = int(idx.slot_ptr[lo]), int(idx.slot_ptr[hi])
        fake = Pairs(self.fake_pop, self.fake_gen, self.fake_row, n=s1 - s0, off=s0)
        mc = int(cand_len[lo:hi].max()) if hi > lo else 0
        samp = cabi.ltg_sample_inputs(hi - lo, max(1, mc), _ptr(self.cand_ptr, lo), _ptr(self.cand_idx), _ptr(self.pop_ptr, lo),
                                      _ptr(self.pop_idx), _ptr(self.n_sample, lo), _ptr(self.slot_ptr, lo), _ptr(self.valid_item),
                                      0, None, None, None)
        return dict(batch=batch, real=real, fake=fake, samp=samp, n_real=r1 - r0, n_slots=s1 - s0, lo=lo, hi=hi, slot0=s0)

    def view(self, b):
        return self._views[b]

    def tower_chunks(self, max_pairs):
        """the epoch's fake-pair slots cut at batch boundaries into runs of at most max_pairs slots (a single batch may exceed it):
        inputs of Engine.fake_tower_batched -- (Pairs over the run, seg_of = batch of each slot, seg_row0 = first slot of every
        batch RELATIVE to the run, first slot of the run)"""
        if self._towers is None:
            slot0 = np.array([v["slot0"] for v in self._views] + [self.n_slots], np.int64)
            seg_of = np.repeat(np.arange(self.n_batches, dtype=np.int32), np.diff(slot0))
            self._seg_of = torch.from_numpy(seg_of if len(seg_of) else np.zeros(1, np.int32)).to(self.device)
            out, b0 = [], 0
            while b0 < self.n_batches:
                b1 = b0 + 1
                while b1 < self.n_batches and slot0[b1 + 1] - slot0[b0] <= max_pairs:
                    b1 += 1
                s0, s1 = int(slot0[b0]), int(slot0[b1])
                if s1 > s0:
                    row0 = torch.from_numpy((slot0[:-1] - s0).astype(np.int32)).to(self.device)
                    out.append(dict(fake=Pairs(self.fake_pop, self.fake_gen, None, n=s1 - s0, off=s0), seg_of=self._seg_of, seg_off=s0, seg_row0=row0, s0=s0))
                b0 = b1
            self._towers = out
        return self._towers

    def span(self, b0, b1):
        """batches [b0, b1) as ONE forward / sampler input (phase C needs no weight update between batches): the CSR rows and
        the sampler's per-user arrays of users [b0 BS, min(N, b1 BS))"""
        key = (b0, b1)
        if key not in self._spans:
            lo, hi = b0 * self.BS, min(self.N, b1 * self.BS)
            batch = CsrRows(self.indptr, self.indices, lo, hi, values=self.values, row_norm2=self.row_norm2)
            cl = np.diff(self.idx.cand_ptr[lo:hi + 1])
            samp = cabi.ltg_sample_inputs(hi - lo, max(1, int(cl.max()) if hi > lo else 1), _ptr(self.cand_ptr, lo), _ptr(self.cand_idx),
                                          _ptr(self.pop_ptr, lo), _ptr(self.pop_idx), _ptr(self.n_sample, lo), _ptr(self.slot_ptr, lo),
                                          _ptr(self.valid_item), 0, None, None, None, self.BS, 0)
            self._spans[key] = dict(batch=batch, samp=samp, lo=lo, hi=hi)
        return self._spans[key]


class EvalData:
    """Fold-in / held-out CSR pair on the device (load_tr_te_data, train.py:83-84, test.py:68-69)."""

    def __init__(self, tr_csr, te_csr, device, item_lo=0, item_hi=None):
        """item_lo/item_hi: this rank's item slab -- the fold-in matrix is cut to those columns (re-indexed from 0, it
        is both the generator input and the -inf mask); the held-out matrix keeps GLOBAL ids on every rank."""
        dev = torch.device(device)
        tr = tr_csr.tocsr()
        te = te_csr.tocsr()
        self.row_norm2 = None
        if (item_lo, item_hi) not in ((0, None), (0, tr.shape[1])):
            self.row_norm2 = torch.from_numpy(np.asarray(tr.multiply(tr).sum(axis=1), dtype=np.float32).reshape(-1)).to(dev)
            tr = tr[:, item_lo:item_hi].tocsr()
        tr.sort_indices()
        te.sort_indices()
        self.n = tr.shape[0]
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)
        self.tr_indptr, self.tr_indices = up(tr.indptr), up(tr.indices)
        self.te_indptr, self.te_indices = up(te.indptr), up(te.indices)
        self.tr_host, self.te_host = tr, te

    def rows(self, lo, hi):
        return (CsrRows(self.tr_indptr, self.tr_indices, lo, hi, row_norm2=self.row_norm2),
                CsrRows(self.te_indptr, self.te_indices, lo, hi))
