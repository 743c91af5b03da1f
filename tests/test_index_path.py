"""Bit-exact index path: ltgan.data_processing (the build's restatement of the reference's
Codes/data_processing.py) against the golden outputs captured by importing the reference itself
(tests/golden/make_golden.py -> askubuntu_golden.npz), on the Askubuntu_Sample files rebuilt from
askubuntu_raw.npz."""
import hashlib
import os

import numpy as np
import pytest

from ltgan import data_processing as dp
from ltgan.dataset import materialize_askubuntu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ds(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("askubuntu"))
    materialize_askubuntu(os.path.join(G, "askubuntu_raw.npz"), d)
    return d


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "askubuntu_golden.npz"))


@pytest.fixture(scope="module")
def loaded(ds):
    n_items = sum(1 for _ in open(os.path.join(ds, "unique_item_id.txt")))
    show2id, present, niche, all_tags, other = dp.load_pop_niche_tags(os.path.join(ds, "item2id.txt"), os.path.join(ds, "item_list.txt"),
                                                                    os.path.join(ds, "niche_items.txt"), n_items)
    fdict, flen, farr = dp.load_item_one_hot_features(os.path.join(ds, "item_list.txt"), show2id, n_items)
    train, uid0 = dp.load_train_data(os.path.join(ds, "train_GAN.csv"), n_items)
    upop = dp.load_user_items(os.path.join(ds, "train_GAN_popular.csv"))
    unic = dp.load_user_items(os.path.join(ds, "train_GAN_niche.csv"))
    oc = dp.load_overlap_coeff(os.path.join(ds, "item2id.txt"), os.path.join(ds, "item_counts.csv"))
    return dict(n_items=n_items, show2id=show2id, present=present, niche=niche, other=other, fdict=fdict, flen=flen, farr=farr,
                train=train, uid0=uid0, upop=upop, unic=unic, oc=oc)


def _ragged(d, keys):
    ptr = np.zeros(len(keys) + 1, np.int64)
    vals = []
    for i, k in enumerate(keys):
        vals += list(d[k])
        ptr[i + 1] = len(vals)
    return ptr, np.asarray(vals, np.int64)


def test_tags_and_features(loaded, gold):
    assert loaded["n_items"] == int(gold["n_items"]) == 1000
    assert len(loaded["show2id"]) == int(gold["n_show2id"])
    assert np.array_equal(np.array(sorted(int(x) for x in loaded["present"])), gold["ids_present"])
    assert np.array_equal(np.array(sorted(loaded["niche"])), gold["niche_tags"]) and len(loaded["niche"]) == 897
    assert np.array_equal(loaded["other"], gold["other_tags"])
    assert np.array_equal(np.array(sorted(loaded["fdict"])), gold["valid_ids"])
    assert not {418, 447, 595} & set(loaded["fdict"])                       # Q9
    assert loaded["flen"] == int(gold["feature_len"]) == 1000
    assert tuple(loaded["farr"].shape) == tuple(gold["feature_arr_shape"])


def test_train_matrix(loaded, gold, ds):
    tr = loaded["train"]
    assert tr.shape == (10001, 1000) and tr.dtype == np.float32 and int(loaded["uid0"]) == int(gold["uid_start_idx"]) == 0
    tr.sort_indices()
    assert np.array_equal(tr.indptr, gold["train_indptr"]) and np.array_equal(tr.indices, gold["train_indices"])
    assert float(tr.data.sum()) == float(gold["train_data_sum"])
    vtr, vte, v0 = dp.load_tr_te_data(os.path.join(ds, "validation_tr.csv"), os.path.join(ds, "validation_te.csv"), 1000)
    assert int(v0) == int(gold["vad_uid_start_idx"]) and vtr.dtype == np.float64
    vtr.sort_indices(); vte.sort_indices()
    assert np.array_equal(vtr.indptr, gold["vtr_indptr"]) and np.array_equal(vtr.indices, gold["vtr_indices"])
    assert np.array_equal(vte.indptr, gold["vte_indptr"]) and np.array_equal(vte.indices, gold["vte_indices"])


def test_user_lists(loaded, gold):
    for name, d in (("pop", loaded["upop"]), ("nic", loaded["unic"])):
        keys = sorted(d)
        assert np.array_equal(np.array(keys), gold[name + "_users"])
        ptr, idx = _ragged(d, keys)
        assert np.array_equal(ptr, gold[name + "_ptr"]) and np.array_equal(idx, gold[name + "_idx"])
    both = set(loaded["upop"]) & set(loaded["unic"])
    assert len(loaded["upop"]) == 9962 and len(loaded["unic"]) == 9572 and len(both) == 9533


def test_overlap_coefficients_bit_exact(loaded, gold):
    oc = loaded["oc"].matrix
    assert oc.shape == (1000, 1000) and oc.dtype == np.float64
    assert hashlib.sha256(np.ascontiguousarray(oc).tobytes()).hexdigest() == str(gold["oc_sha256"])
    assert loaded["oc"][0][1] == float(gold["oc_0_1"]) == 0.26046511627906976 and loaded["oc"][0][0] == 1.0


def test_real_pairs_bit_exact(loaded, gold):
    xn, xp = dp.load_vectors(loaded["upop"], loaded["unic"], loaded["oc"], loaded["fdict"], 10001)
    keys = sorted(xn)
    assert np.array_equal(np.array(keys), gold["vec_users"])
    ptr, nic = _ragged(xn, keys)
    _, pop = _ragged(xp, keys)
    assert np.array_equal(ptr, gold["vec_ptr"]) and np.array_equal(nic, gold["vec_niche"]) and np.array_equal(pop, gold["vec_pop"])
    assert len(nic) == 92814


def test_candidate_sets_bit_exact(loaded, gold):
    cand = dp.load_items_to_sample(loaded["upop"], loaded["unic"], loaded["niche"], loaded["oc"], 10001)
    keys = sorted(cand)
    assert np.array_equal(np.array(keys), gold["cand_users"])
    ptr, idx = _ragged(cand, keys)
    assert np.array_equal(ptr, gold["cand_ptr"]) and np.array_equal(idx, gold["cand_idx"])
    lens = np.diff(ptr)
    assert lens.min() == 10 and lens.max() == 900


def test_sparse_overlap_form_is_bit_exact(ds, loaded, gold):
    """the co-occurrence CSR form (no I x I table) reproduces the same real pairs, candidate sets and
    coefficients bit for bit on Askubuntu_Sample."""
    sp = dp.load_overlap_coeff(os.path.join(ds, "item2id.txt"), os.path.join(ds, "item_counts.csv"), sparse_form=True)
    assert isinstance(sp, dp._SparseOverlapView) and sp.sparse.inter.nnz < 1000 * 1000
    dense = loaded["oc"].matrix
    rs = np.random.default_rng(0)
    rows, cols = rs.integers(0, 1000, 40), rs.integers(0, 1000, 60)
    assert np.array_equal(sp.sparse.block(rows, cols), dense[np.ix_(rows, cols)], equal_nan=True)
    assert sp[0][1] == float(gold["oc_0_1"]) and sp[0][0] == 1.0 and len(sp) == len(loaded["oc"])
    xn, xp = dp.load_vectors(loaded["upop"], loaded["unic"], sp, loaded["fdict"], 10001)
    keys = sorted(xn)
    ptr, nic = _ragged(xn, keys)
    _, pop = _ragged(xp, keys)
    assert np.array_equal(ptr, gold["vec_ptr"]) and np.array_equal(nic, gold["vec_niche"]) and np.array_equal(pop, gold["vec_pop"])
    cand = dp.load_items_to_sample(loaded["upop"], loaded["unic"], loaded["niche"], sp, 10001)
    keys = sorted(cand)
    ptr, idx = _ragged(cand, keys)
    assert np.array_equal(np.array(keys), gold["cand_users"])
    assert np.array_equal(ptr, gold["cand_ptr"]) and np.array_equal(idx, gold["cand_idx"])


def test_valid_item_ids_equal_the_one_hot_dict_keys(tmp_path):
    """IndexData.from_dir no longer builds the I x I one-hot table: load_valid_item_ids must give exactly the key set of
    load_item_one_hot_features (data_processing.py:54-59), 997 ids with 418 / 447 / 595 missing (Q9)."""
    from ltgan import data_processing as dp
    from ltgan.dataset import materialize_askubuntu
    d = str(tmp_path / "ds")
    materialize_askubuntu(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "askubuntu_raw.npz"), d)
    show2id, *_ = dp.load_pop_niche_tags(os.path.join(d, "item2id.txt"), os.path.join(d, "item_list.txt"), os.path.join(d, "niche_items.txt"), 1000)
    fdict, flen, _ = dp.load_item_one_hot_features(os.path.join(d, "item_list.txt"), show2id, 1000)
    ids = dp.load_valid_item_ids(os.path.join(d, "item_list.txt"), show2id)
    assert ids == set(fdict.keys()) and len(ids) == 997 and not ({418, 447, 595} & ids) and flen == 1000


def test_set_difference_order_predicate_matches_cpython():
    """_cpython_set_difference_is_sorted may only say True when `so - other` really iterates in ascending order on this
    interpreter (the candidate ranking's tie order, SURVEY 8/c4b, then follows without performing the set operation)."""
    rng = np.random.default_rng(0)
    said_true = 0
    for trial in range(300):
        top = int(rng.choice([50, 1000, 5000, 40000]))
        n = int(rng.integers(1, min(top, 6000)))
        so = set(rng.choice(top, size=n, replace=False).tolist())
        n_other = int(rng.integers(0, max(1, n // 2)))
        other = set(rng.choice(sorted(so), size=min(n_other, n), replace=False).tolist()) if n_other else set()
        if dp._cpython_set_difference_is_sorted(np.asarray(sorted(so)), len(other)):
            said_true += 1
            res = list(so - other)
            assert res == sorted(res), (top, n, len(other))
    assert said_true > 50


def test_sparse_overlap_form_above_the_switch_is_bit_exact(tmp_path):
    """SURVEY 8/f3 at a tag count above SPARSE_OVERLAP_MIN_TAGS: a synthetic dataset DIRECTORY in the reference's file
    formats (9 000 tags) through the whole index path -- the co-occurrence CSR form (what load_overlap_coeff picks by itself
    at this size; per-user work proportional to the co-occurring tags) against the dense I x I table: real pairs and
    candidate sets identical, and IndexData.from_dir ingests the directory."""
    from ltgan.dataset import IndexData
    from ltgan.synthetic import write_dataset_dir
    d = str(tmp_path / "syn9000")
    I = 9000
    assert I > dp.SPARSE_OVERLAP_MIN_TAGS
    write_dataset_dir(d, n_items=I, n_users=1200, n_eval_users=60)
    j = lambda n: os.path.join(d, n)
    show2id, present, niche, _, _ = dp.load_pop_niche_tags(j("item2id.txt"), j("item_list.txt"), j("niche_items.txt"), I)
    upop, unic = dp.load_user_items(j("train_GAN_popular.csv")), dp.load_user_items(j("train_GAN_niche.csv"))
    train, _ = dp.load_train_data(j("train_GAN.csv"), I)
    N = train.shape[0]
    valid = dp.load_valid_item_ids(j("item_list.txt"), show2id)
    assert len(valid) == I - 2 and 7 not in valid and 4242 not in valid
    dense = dp.load_overlap_coeff(j("item2id.txt"), j("item_counts.csv"), sparse_form=False)
    auto = dp.load_overlap_coeff(j("item2id.txt"), j("item_counts.csv"))
    assert isinstance(auto, dp._SparseOverlapView) and isinstance(dense, dp._OverlapView)
    a, b = dp.load_vectors(upop, unic, dense, valid, N), dp.load_vectors(upop, unic, auto, valid, N)
    assert a[0] == b[0] and a[1] == b[1] and sum(len(v) for v in a[0].values()) > 3000
    ca, cb = dp.load_items_to_sample(upop, unic, niche, dense, N), dp.load_items_to_sample(upop, unic, niche, auto, N)
    assert ca.keys() == cb.keys() and all(np.array_equal(ca[k], cb[k]) for k in ca) and len(ca) > 900
    idx = IndexData.from_dir(d)
    assert idx.n_items == I and idx.N == N and int(idx.valid_item.sum()) == I - 2 and len(idx.cand_idx) == sum(len(v) for v in cb.values())
