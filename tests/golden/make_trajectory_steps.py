#!/usr/bin/env python3
"""Generates tests/golden/oracle_trajectory_steps.npz: SURVEY 8/d6's per-step gate -- the fp64 oracle (oracle/trajectory.py) on Askubuntu_Sample
with config.ini's schedule (S = NUM_EPOCH / 8 = 10 sub-epochs, BATCH_SIZE 100; Codes/train.py:287-329): d_loss of EACH of the first N_STEPS
discriminator updates of global epoch 0, then -- behind the WHOLE discriminator phase (1 010 updates) -- the loss triplet of EACH of the first
N_STEPS generator updates.  Same initial weights, counter-RNG streams and call sequence as ltgan.trainer.Trainer.  Pure oracle output (no
reference source, no GPU).

usage: python tests/golden/make_trajectory_steps.py   (about 6 minutes on 8 cores: 101 creation forwards + 1 010 D steps + 50 G steps in fp64)
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ltgan.dataset import IndexData, materialize_askubuntu  # noqa: E402
from oracle import ltg_oracle as O  # noqa: E402
from oracle.trajectory import OracleTrainer  # noqa: E402

S, HS, LR, SEED, N_STEPS = 10, (100, 150, 250, 300), 1e-4, 98765, 50

if __name__ == "__main__":
    d = tempfile.mkdtemp()
    materialize_askubuntu(os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz"), d)
    idx = IndexData.from_dir(d)
    P = O.init_generator(idx.n_items, seed=7)
    D = O.init_discriminator(idx.n_items, *HS, seed=8)
    tr = OracleTrainer(idx, P, D, HS, lr=LR, S=S, seed=SEED, quant=True, shuffle_seed=0)
    t0 = time.time()
    tr.create_phase()
    out = dict(S=S, hs=np.array(HS), lr=LR, seed=SEED, gen_seed=7, disc_seed=8, n_steps=N_STEPS)
    out["cnt"] = np.array([len(tr.fake[b][0]) for b in range(tr.n_batches)])
    out["order"] = tr.order.copy()
    out["fake_gen"] = np.concatenate([tr.fake[b][1] for b in range(tr.n_batches)]).astype(np.int16)
    print("creation done, %.0fs" % (time.time() - t0), flush=True)
    sub = tr.d_phase()
    out["d_loss_steps"] = np.array(tr.d_log[:N_STEPS])
    out["d_loss_sub_epochs"] = np.array(sub)
    out["d_steps_total"] = len(tr.d_log)
    print("D phase done (%d steps), %.0fs" % (len(tr.d_log), time.time() - t0), flush=True)
    tr.g_phase(max_steps=N_STEPS)
    out["g_loss_steps"] = np.array(tr.g_log[:N_STEPS])          # [n, 4]: g_loss, vae_loss, gan_loss, anneal
    print("first %d G steps done, %.0fs" % (N_STEPS, time.time() - t0), flush=True)
    print("d", out["d_loss_steps"][:5], "g", out["g_loss_steps"][:3])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trajectory_steps.npz"), **out)
