#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's own pure-numpy
modules read-only.  Runs only in the build container (needs /root/reference); the
fixtures it writes are data (inputs + expected outputs), never reference source.

  askubuntu_raw.npz     the sample dataset's files as integer arrays (so the index path can be
                        re-run anywhere; the loaders' inputs)
  askubuntu_golden.npz  outputs of data_processing.py's loaders on it (the bit-exact index path)
  sampler_golden.npz    sample.py:sample_from_generator_new under fixed np.random seeds
  metrics_golden.npz    eval_functions.py NDCG/Recall on seeded predictions (run under
                        /opt/conda/bin/python3.9 where bottleneck exists, tensorflow stubbed)

usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd

sys.dont_write_bytecode = True
REF = "/root/reference"
DS = os.path.join(REF, "Dataset", "Askubuntu_Sample")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "Codes"))


def ragged(d, keys):
    ptr = np.zeros(len(keys) + 1, np.int64)
    vals = []
    for i, k in enumerate(keys):
        v = list(d[k])
        vals += v
        ptr[i + 1] = ptr[i] + len(v)
    return ptr, np.asarray(vals, np.int64)


def raw():
    out = {}
    for name in ["train_GAN", "train_GAN_popular", "train_GAN_niche", "validation_tr", "validation_te", "test_tr", "test_te"]:
        tp = pd.read_csv(os.path.join(DS, name + ".csv"))
        out[name + "_uid"] = tp["uid"].to_numpy(np.int32)
        out[name + "_sid"] = tp["sid"].to_numpy(np.int16)
    ic = pd.read_csv(os.path.join(DS, "item_counts.csv"))
    out["item_counts_userId"] = ic["userId"].to_numpy(np.int32)
    out["item_counts_tagId"] = ic["tagId"].to_numpy(np.int32)
    out["item_counts_rating"] = ic["rating"].to_numpy(np.float32)
    i2i = pd.read_csv(os.path.join(DS, "item2id.txt"), sep="\t", header=None)
    out["item2id_raw"] = i2i[0].to_numpy(np.int32)
    out["item2id_id"] = i2i[1].to_numpy(np.int32)
    out["item_list"] = np.loadtxt(os.path.join(DS, "item_list.txt"), dtype=np.int32)
    out["niche_items"] = np.loadtxt(os.path.join(DS, "niche_items.txt"), dtype=np.int32)
    out["unique_item_id"] = np.loadtxt(os.path.join(DS, "unique_item_id.txt"), dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "askubuntu_raw.npz"), **out)
    return out


def index_path():
    import data_processing as dp  # the reference module, unmodified
    n_items = sum(1 for _ in open(os.path.join(DS, "unique_item_id.txt")))
    SHOW2ID, IDs_present, NICHE, ALL, OTHER = dp.load_pop_niche_tags(os.path.join(DS, "item2id.txt"), os.path.join(DS, "item_list.txt"),
                                                                   os.path.join(DS, "niche_items.txt"), n_items)
    FDICT, FLEN, FARR = dp.load_item_one_hot_features(os.path.join(DS, "item_list.txt"), SHOW2ID, n_items)
    train, uid0 = dp.load_train_data(os.path.join(DS, "train_GAN.csv"), n_items)
    vtr, vte, vuid0 = dp.load_tr_te_data(os.path.join(DS, "validation_tr.csv"), os.path.join(DS, "validation_te.csv"), n_items)
    upop = dp.load_user_items(os.path.join(DS, "train_GAN_popular.csv"))
    unic = dp.load_user_items(os.path.join(DS, "train_GAN_niche.csv"))
    OC = dp.load_overlap_coeff(os.path.join(DS, "item2id.txt"), os.path.join(DS, "item_counts.csv"))
    N = train.shape[0]
    xn, xp = dp.load_vectors(upop, unic, OC, FDICT, N)
    cand = dp.load_items_to_sample(upop, unic, NICHE, OC, N)
    ocm = np.array([[OC[a][b] for b in range(n_items)] for a in range(n_items)], np.float64)
    out = dict(n_items=n_items, N=N, uid_start_idx=int(uid0), vad_uid_start_idx=int(vuid0),
               n_show2id=len(SHOW2ID), ids_present=np.array(sorted(int(x) for x in IDs_present), np.int32),
               niche_tags=np.array(sorted(NICHE), np.int32), other_tags=np.asarray(OTHER, np.int32),
               valid_ids=np.array(sorted(FDICT.keys()), np.int32), feature_len=FLEN, feature_arr_shape=np.array(FARR.shape),
               train_indptr=train.indptr.astype(np.int32), train_indices=train.indices.astype(np.int16),
               train_data_sum=float(train.data.sum()),
               vtr_indptr=vtr.indptr.astype(np.int32), vtr_indices=vtr.indices.astype(np.int16),
               vte_indptr=vte.indptr.astype(np.int32), vte_indices=vte.indices.astype(np.int16),
               oc_sha256=hashlib.sha256(ocm.tobytes()).hexdigest(), oc_row0=ocm[0], oc_diag=np.diag(ocm).copy(),
               oc_0_1=ocm[0, 1])
    keys_pop = sorted(upop)
    keys_nic = sorted(unic)
    out["pop_users"] = np.array(keys_pop, np.int32)
    out["pop_ptr"], out["pop_idx"] = ragged(upop, keys_pop)
    out["nic_users"] = np.array(keys_nic, np.int32)
    out["nic_ptr"], out["nic_idx"] = ragged(unic, keys_nic)
    kv = sorted(xn)
    out["vec_users"] = np.array(kv, np.int32)
    out["vec_ptr"], out["vec_niche"] = ragged(xn, kv)
    _, out["vec_pop"] = ragged(xp, kv)
    kc = sorted(cand)
    out["cand_users"] = np.array(kc, np.int32)
    out["cand_ptr"], out["cand_idx"] = ragged(cand, kc)
    for k in ("pop_idx", "nic_idx", "vec_niche", "vec_pop", "cand_idx"):
        out[k] = out[k].astype(np.int16)
    for k in ("pop_ptr", "nic_ptr", "vec_ptr", "cand_ptr"):
        out[k] = out[k].astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "askubuntu_golden.npz"), **out)
    print("index path:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k not in ("oc_row0", "oc_diag")})
    return out


def sampler():
    import sample as ref_sample  # the reference module, unmodified
    rng = np.random.default_rng(2024)
    cases = []
    for case in range(40):
        n_items = 1000
        nc = int(rng.integers(10, 120))
        cand = np.sort(rng.choice(n_items, nc, replace=False))
        p = rng.dirichlet(np.ones(nc) * 0.3)
        if case % 5 == 0:  # some exact zeros: exercises the exception-driven decrement (Q10)
            p[rng.random(nc) < 0.6] = 0.0
            if p.sum() == 0:
                p[0] = 1.0
        k = int(rng.integers(1, max(2, nc // 2)))
        seed = 1000 + case
        np.random.seed(seed)
        binv, ids = ref_sample.sample_from_generator_new(cand, p.astype(np.float32), k, n_items)
        cases.append(dict(cand=cand, p=p.astype(np.float32), k=k, seed=seed, ids=np.asarray(ids), bin_nnz=int(binv.sum())))
    out = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out["c%d_%s" % (i, k)] = v
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "sampler_golden.npz"), **out)
    print("sampler cases:", len(cases))


METRIC_SNIPPET = r'''
import sys, types, json
sys.dont_write_bytecode = True
sys.modules['tensorflow'] = types.ModuleType('tensorflow')   # imported at eval_functions.py:1, never used
sys.path.insert(0, '/root/reference/Codes')
import numpy as np
from scipy import sparse
import eval_functions as ef
res = {}
for seed, (n, I, dens) in enumerate([(64, 1000, 0.01), (33, 257, 0.05), (16, 1000, 0.002)]):
    rs = np.random.RandomState(seed)
    pred = rs.rand(n, I).astype(np.float32)
    held = sparse.csr_matrix((rs.rand(n, I) < dens).astype(np.float64))
    tr = (rs.rand(n, I) < 0.02)
    tr[held.toarray() > 0] = False
    pred[tr] = -np.inf                                         # train.py:341
    nd = ef.NDCG_binary_at_k_batch(pred, held, k=100)
    r20, _ = ef.Recall_at_k_batch(pred, held, k=20)
    r50, _ = ef.Recall_at_k_batch(pred, held, k=50)
    res[str(seed)] = dict(n=n, I=I, dens=dens, ndcg=[float(x) for x in nd], r20=[float(x) for x in r20], r50=[float(x) for x in r50])
print(json.dumps(res))
'''


def metrics():
    out = subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", "-c", METRIC_SNIPPET], capture_output=True, text=True, cwd="/tmp")
    if out.returncode != 0:
        raise RuntimeError(out.stderr)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    flat = {}
    for seed, r in res.items():
        flat["s%s_shape" % seed] = np.array([r["n"], r["I"]])
        flat["s%s_dens" % seed] = r["dens"]
        for k in ("ndcg", "r20", "r50"):
            flat["s%s_%s" % (seed, k)] = np.array(r[k], np.float64)
    flat["n_cases"] = len(res)
    np.savez_compressed(os.path.join(OUT, "metrics_golden.npz"), **flat)
    print("metrics:", {s: (np.mean(r["ndcg"]), np.mean(r["r20"]), np.mean(r["r50"])) for s, r in res.items()})


if __name__ == "__main__":
    os.chdir("/tmp")
    raw()
    sampler()
    metrics()
    index_path()
