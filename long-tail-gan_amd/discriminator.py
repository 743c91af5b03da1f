"""Discriminator factory of the reference (Codes/discriminator.py:3-58):

    y_data, y_generated, d_params, x_generated_id, x_popular_n_id, x_popular_g_id, x_niche_id, \
        item_feature_arr, keep_prob = discriminator(n_items, FEATURE_LEN, h0, h1, h2, h3)

Returns symbolic handles with the reference's names; `d_params` = [w1, b1, w2, b2, w3, b3, w4, b4]
device tensors (discriminator.py:47).  The item embedding table (discriminator.py:14) is frozen (Q4).
"""
from __future__ import annotations

from .generator import Fetch, Placeholder


def discriminator(n_items, FEATURE_LEN, h0_size, h1_size, h2_size, h3_size, engine=None):
    if engine is None:
        raise ValueError("pass engine=<the Engine created by generator_VAECF> (one shared graph, train.py:127-136)")
    want = (h0_size, h1_size, h2_size, h3_size)
    have = (engine.h0, engine.h1, engine.h2, engine.h3)
    if want != have or engine.feature_len != FEATURE_LEN or engine.I_global != n_items:
        raise ValueError("engine was built for D sizes %s / feature_len %d / n_items %d" % (have, engine.feature_len, engine.I_global))
    x_generated_id = Placeholder("x_generated")      # discriminator.py:5
    x_popular_n_id = Placeholder("x_popular_n")      # :6
    x_popular_g_id = Placeholder("x_popular_g")      # :7
    x_niche_id = Placeholder("x_niche")              # :8
    item_feature_arr = Placeholder("item_feature_arr")  # :10 -- dead in the reference (Q7), accepted and ignored
    keep_prob = Placeholder("keep_prob")             # :12
    return (Fetch("y_data"), Fetch("y_generated"), engine.discriminator_params_tf(), x_generated_id, x_popular_n_id,
            x_popular_g_id, x_niche_id, item_feature_arr, keep_prob)
