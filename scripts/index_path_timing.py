#!/usr/bin/env python3
"""Timings of the index path (ltgan.data_processing) on a synthetic dataset directory of ML-20M's tag count
(20 000 tags): the co-occurrence CSR form that load_overlap_coeff picks above 8 192 tags.
usage: python scripts/index_path_timing.py [n_items] [n_users]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltgan import data_processing as dp  # noqa: E402
from ltgan.synthetic import write_dataset_dir  # noqa: E402

I = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
U = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
d = tempfile.mkdtemp()
t = time.time(); write_dataset_dir(d, n_items=I, n_users=U, n_eval_users=200); print("write_dataset_dir      %7.2f s" % (time.time() - t), flush=True)
j = lambda n: os.path.join(d, n)
t = time.time(); show2id, present, niche, _, _ = dp.load_pop_niche_tags(j("item2id.txt"), j("item_list.txt"), j("niche_items.txt"), I); print("load_pop_niche_tags    %7.2f s" % (time.time() - t))
t = time.time(); upop, unic = dp.load_user_items(j("train_GAN_popular.csv")), dp.load_user_items(j("train_GAN_niche.csv")); print("load_user_items x2     %7.2f s" % (time.time() - t))
t = time.time(); train, _ = dp.load_train_data(j("train_GAN.csv"), I); N = train.shape[0]; print("load_train_data        %7.2f s  (%d users, %d interactions)" % (time.time() - t, N, train.nnz))
valid = dp.load_valid_item_ids(j("item_list.txt"), show2id)
t = time.time(); oc = dp.load_overlap_coeff(j("item2id.txt"), j("item_counts.csv")); print("load_overlap_coeff     %7.2f s  (%s, %d stored co-occurrences)" % (time.time() - t, type(oc).__name__, oc.sparse.inter.nnz))
t = time.time(); xn, xp = dp.load_vectors(upop, unic, oc, valid, N); print("load_vectors           %7.2f s  (%d real pairs)" % (time.time() - t, sum(len(v) for v in xn.values())))
t = time.time(); cand = dp.load_items_to_sample(upop, unic, niche, oc, N); print("load_items_to_sample   %7.2f s  (%d users with candidates)" % (time.time() - t, len(cand)))
