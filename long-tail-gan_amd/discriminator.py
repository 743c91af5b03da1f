"""Discriminator factory of the reference (Codes/discriminator.py:3-58):

    y_data, y_generated, d_params, x_generated_id, x_popular_n_id, x_popular_g_id, x_niche_id, \
        item_feature_arr, keep_prob = discriminator(n_items, FEATURE_LEN, h0, h1, h2, h3)

Returns symbolic handles with the reference's names; `d_params` = [w1, b1, w2, b2, w3, b3, w4, b4]
device tensors (discriminator.py:47).  The item embedding table (discriminator.py:14) is frozen (Q4).
"""
from __future__ import annotations

from .generator import Fetch, Placeholder, current_engine


def discriminator(n_items, FEATURE_LEN, h0_size, h1_size, h2_size, h3_size, engine=None):
    """The reference's six positional arguments (train.py:136, test.py:85).  The discriminator lives in the engine of the
    preceding generator(...) call (one shared graph, train.py:127-136); when that engine was built for other layer sizes
    -- generator(pro_dir) only knows ./config.ini -- its discriminator part is re-created with the requested ones."""
    if engine is None:
        engine = current_engine()
    if engine is None:
        raise ValueError("call generator(pro_dir) first: the discriminator is built into the generator's engine (train.py:130-136)")
    if engine.I_global != n_items:
        raise ValueError("the current engine was built for %d items, discriminator() was asked for %d" % (engine.I_global, n_items))
    want = (int(h0_size), int(h1_size), int(h2_size), int(h3_size))
    if want != (engine.h0, engine.h1, engine.h2, engine.h3) or engine.feature_len != FEATURE_LEN:
        engine.resize_discriminator(want, FEATURE_LEN)
    x_generated_id = Placeholder("x_generated")      # discriminator.py:5
    x_popular_n_id = Placeholder("x_popular_n")      # :6
    x_popular_g_id = Placeholder("x_popular_g")      # :7
    x_niche_id = Placeholder("x_niche")              # :8
    item_feature_arr = Placeholder("item_feature_arr")  # :10 -- dead in the reference (Q7), accepted and ignored
    keep_prob = Placeholder("keep_prob")             # :12
    return (Fetch("y_data"), Fetch("y_generated"), engine.discriminator_params_tf(), x_generated_id, x_popular_n_id,
            x_popular_g_id, x_niche_id, item_feature_arr, keep_prob)
