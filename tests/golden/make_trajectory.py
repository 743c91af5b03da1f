#!/usr/bin/env python3
"""Generates tests/golden/oracle_trajectory.npz: the fp64 oracle trained on Askubuntu_Sample for
EPOCHS global epochs x S sub-epochs with the shared counter RNG (oracle/trajectory.py), bf16-quantised
decoder operands like the default HIP path.  Pure oracle output (no reference source, no GPU).

usage: python tests/golden/make_trajectory.py   (about 3-4 minutes on 8 cores)
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ltgan import data_processing as dp  # noqa: E402
from ltgan.dataset import IndexData, materialize_askubuntu  # noqa: E402
from oracle import ltg_oracle as O  # noqa: E402
from oracle.trajectory import OracleTrainer  # noqa: E402

EPOCHS, S, HS, LR, SEED = 2, 2, (100, 150, 250, 300), 1e-4, 98765

if __name__ == "__main__":
    d = tempfile.mkdtemp()
    materialize_askubuntu(os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz"), d)
    idx = IndexData.from_dir(d)
    vtr, vte, _ = dp.load_tr_te_data(os.path.join(d, "validation_tr.csv"), os.path.join(d, "validation_te.csv"), idx.n_items)
    P = O.init_generator(idx.n_items, seed=7)
    D = O.init_discriminator(idx.n_items, *HS, seed=8)
    tr = OracleTrainer(idx, P, D, HS, lr=LR, S=S, seed=SEED, quant=True, shuffle_seed=0)
    out = dict(epochs=EPOCHS, S=S, hs=np.array(HS), lr=LR, seed=SEED, gen_seed=7, disc_seed=8)
    t0 = time.time()
    for e in range(EPOCHS):
        tr.create_phase()
        out["e%d_cnt" % e] = np.array([len(tr.fake[b][0]) for b in range(tr.n_batches)])
        out["e%d_order" % e] = tr.order.copy()
        out["e%d_fake_gen" % e] = np.concatenate([tr.fake[b][1] for b in range(tr.n_batches)]).astype(np.int16)
        out["e%d_d_loss" % e] = np.array(tr.d_phase())
        out["e%d_g_loss" % e] = np.array(tr.g_phase())
        out["e%d_metrics" % e] = np.array(tr.validate(vtr, vte, rng_step=1000 + e))
        print("epoch", e, "d", out["e%d_d_loss" % e], "g", out["e%d_g_loss" % e][-1], "ndcg/r20/r50", out["e%d_metrics" % e],
              "%.0fs" % (time.time() - t0), flush=True)
    out["final_bp1_head"] = tr.P["bp1"][:64]
    out["final_w4_head"] = tr.D["w4"][:64, 0]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trajectory.npz"), **out)
