"""The reference's own loop shape (Codes/train.py:127-346) over the Session facade: graph handles from the wrapper
factories, the five sess.run signatures with the reference's dense feeds, NumPy sampling in between.  Checked against a
twin Engine driven directly through the C ABI with the same counters (bit-identical) and against the oracle."""
import os

import numpy as np
import pytest

from oracle import ltg_oracle as O
from oracle.cpu_port import sample_from_generator_new
import helpers as Hh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_loop_over_session(tmp_path, monkeypatch):
    import torch
    from ltgan.data_processing import (load_train_data, load_tr_te_data, load_user_items, load_overlap_coeff, load_pop_niche_tags,
                                       load_items_to_sample, load_vectors)
    from ltgan.data_processing import load_item_one_hot_features as load_item_features
    from ltgan.dataset import materialize_askubuntu, batch_csc
    from ltgan.discriminator import discriminator
    from ltgan.engine import CsrRows, Engine, Pairs
    from ltgan.generator import generator_VAECF as generator
    from ltgan.session import Session, adversarial_graph

    DATA_DIR = str(tmp_path / "Askubuntu_Sample") + "/"
    materialize_askubuntu(os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz"), DATA_DIR)
    show2id_path, niche_tags_path = DATA_DIR + "item2id.txt", DATA_DIR + "niche_items.txt"
    user_tag_matrix_path, item_list_path, pro_dir = DATA_DIR + "item_counts.csv", DATA_DIR + "item_list.txt", DATA_DIR
    n_items = len(open(pro_dir + "unique_item_id.txt").read().split())
    SHOW2ID, IDs_present, NICHE_TAGS, ALL_TAGS, OTHER_TAGS = load_pop_niche_tags(show2id_path, item_list_path, niche_tags_path, n_items)
    ITEM_FEATURE_DICT, FEATURE_LEN, ITEM_FEATURE_ARR = load_item_features(item_list_path, SHOW2ID, n_items)
    train_data, uid_start_idx = load_train_data(pro_dir + "train_GAN.csv", n_items)
    vad_data_tr, vad_data_te, _ = load_tr_te_data(pro_dir + "validation_tr.csv", pro_dir + "validation_te.csv", n_items)
    user_popular_data = load_user_items(pro_dir + "train_GAN_popular.csv")
    user_niche_data = load_user_items(pro_dir + "train_GAN_niche.csv")
    OVERLAP_COEFFS = load_overlap_coeff(show2id_path, user_tag_matrix_path)
    N = train_data.shape[0]
    user_x_niche_vectors, user_x_popular_n_vectors = load_vectors(user_popular_data, user_niche_data, OVERLAP_COEFFS, ITEM_FEATURE_DICT, N)
    USER_TAGS_TO_SAMPLE = load_items_to_sample(user_popular_data, user_niche_data, NICHE_TAGS, OVERLAP_COEFFS, N)
    h = h0_size, h1_size, h2_size, h3_size = (100, 150, 250, 300)
    LEARNING_RATE = 1e-4
    BATCH_SIZE, NB = 100, 3

    # --- graph (train.py:127-164): the reference's LITERAL argument lists -- `generator(pro_dir)` (train.py:130) and the six
    # positional arguments of `discriminator(...)` (train.py:136); precision is the only thing set from outside the call
    monkeypatch.setenv("LTGAN_PRECISION", "fp32")
    monkeypatch.chdir(tmp_path)              # no config.ini here: the shipped defaults apply
    generator_network, generator_out, g_vae_loss, g_params, p_dims, total_anneal_steps, anneal_cap = generator(pro_dir)
    disc = discriminator(n_items, FEATURE_LEN, h0_size, h1_size, h2_size, h3_size)
    y_data, y_generated, d_params, x_generated_id, x_popular_n_id, x_popular_g_id, x_niche_id, item_feature_arr, keep_prob = disc
    G = adversarial_graph(generator_network, generator_out, g_vae_loss, disc, learning_rate=LEARNING_RATE)
    generated_tags, sampled_cnt, gen_lambda = G.generated_tags, G.sampled_cnt, G.gen_lambda
    sess = Session(generator_network.engine)
    assert [tuple(p.shape) for p in g_params] == [(n_items, 600), (600, 400), (200, 600), (600, n_items), (600,), (400,), (600,), (n_items,)]

    # twin engine with identical parameters, driven directly
    eng = generator_network.engine
    twin = Engine(n_items, h_sizes=h, lr=1e-4, feature_len=FEATURE_LEN, precision="fp32", seed=eng.cfg.seed)
    twin.set_generator([p.cpu().numpy() for p in eng.g_p])
    twin.set_discriminator(eng.d_emb.cpu().numpy(), [p.cpu().numpy() for p in eng.d_p])
    dev = twin.device
    t = lambda a, dt=np.int32: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dt))).to(dev)
    step = 0

    np.random.seed(5)
    cache = []
    for bnum, st_idx in enumerate(range(0, NB * BATCH_SIZE, BATCH_SIZE)):
        end_idx = min(st_idx + BATCH_SIZE, N)
        X = train_data[st_idx:end_idx].toarray().astype("float32")
        curr_generator_out = sess.run(generator_out, feed_dict={generator_network.input_ph: X})
        step += 1
        assert curr_generator_out.shape == X.shape and abs(curr_generator_out.sum(1) - 1).max() < 1e-4
        if bnum == 0:       # oracle check of the first forward (dropout ON at inference: Q3)
            P = Hh.engine_to_gen([p.cpu().numpy() for p in eng.g_p])
            mask = Hh.dropout_mask_dense(int(eng.cfg.seed), 2 * step, X.shape[0], n_items, 0.75)
            F = O.vae_forward(P, X, mask, 0.75, np.zeros((X.shape[0], 200)), 0.0, 1.0, np.float64, quant=False)
            assert np.max(np.abs(curr_generator_out - F["probs"]) / F["probs"]) < 2e-3
        curr_x_popular_n, curr_x_niche, curr_x_popular_g, curr_x_generated = [], [], [], []
        total_sampled_cnt, total_sampled_tags = 0, []
        for ii, user_idx in enumerate(range(st_idx, end_idx)):
            u = user_idx + uid_start_idx
            if u not in user_popular_data or u not in user_niche_data:
                total_sampled_tags.append([0] * n_items)
                continue
            curr_pop_vectors, curr_niche_vectors = user_popular_data[u], user_niche_data[u]
            curr_x_niche += user_x_niche_vectors[u]
            curr_x_popular_n += user_x_popular_n_vectors[u]
            tags_bin, tags = sample_from_generator_new(USER_TAGS_TO_SAMPLE[u], curr_generator_out[ii, USER_TAGS_TO_SAMPLE[u]],
                                                       len(curr_niche_vectors), n_items)
            tags.sort()
            for gid in tags:
                pid = curr_pop_vectors[np.random.choice(range(len(curr_pop_vectors)))]
                if gid not in ITEM_FEATURE_DICT or pid not in ITEM_FEATURE_DICT:
                    tags_bin[gid] = 0
                    continue
                curr_x_generated.append(gid)
                curr_x_popular_g.append(pid)
                total_sampled_cnt += 1
            total_sampled_tags.append(tags_bin)
        assert curr_x_generated
        cache.append((X, np.asarray(total_sampled_tags), np.asarray(curr_x_generated), np.asarray(curr_x_popular_g),
                      np.asarray(curr_x_popular_n), np.asarray(curr_x_niche), total_sampled_cnt))

    for (X, tags, xg, xpg, xpn, xn, cnt) in cache:                      # train.py:300
        _, curr_d_loss = sess.run([G.d_trainer, G.d_loss_mean], feed_dict={
            generator_network.input_ph: X, x_popular_n_id: xpn, x_popular_g_id: xpg, x_niche_id: xn, x_generated_id: xg,
            generated_tags: tags, sampled_cnt: cnt, keep_prob: np.sum(0.7).astype(np.float32), item_feature_arr: ITEM_FEATURE_ARR})
        step += 1
        want = float(twin.d_step(Pairs(t(xpn), t(xn)), Pairs(t(xpg), t(xg)), float(np.float32(0.7)), rng_step=2 * step).cpu()[0])
        assert _ is None and float(curr_d_loss) == np.float32(want)
    update_count = 0.0
    for (X, tags, xg, xpg, xpn, xn, cnt) in cache:                      # train.py:326
        anneal = min(anneal_cap, 1. * update_count / total_anneal_steps)
        update_count += 1
        _, curr_g_loss, t1, t2 = sess.run([G.g_trainer, G.g_loss_mean, g_vae_loss, G.gan_loss], feed_dict={
            generator_network.input_ph: X, x_popular_n_id: xpn, x_popular_g_id: xpg, x_niche_id: xn, x_generated_id: xg,
            generated_tags: tags, sampled_cnt: cnt, generator_network.keep_prob_ph: 0.75, generator_network.is_training_ph: 1,
            generator_network.anneal_ph: anneal, gen_lambda: 1.0, keep_prob: np.sum(0.7).astype(np.float32)})
        step += 1
        import scipy.sparse as sp
        Xs = sp.csr_matrix(X)
        slot, uptr, rowidx, pos = batch_csc(Xs, 0, Xs.shape[0], n_items)
        batch = CsrRows(t(Xs.indptr), t(Xs.indices), 0, Xs.shape[0], slot=t(slot), uptr=t(uptr), rowidx=t(rowidx), csr_pos=t(pos),
                        n_unique=len(uptr) - 1)
        rows = np.nonzero(tags)[0]
        acts = twin.new_acts(Xs.shape[0])
        want = twin.g_step(batch, Pairs(t(xpg), t(xg), t(rows)), acts, torch.tensor([cnt], dtype=torch.int32, device=dev), anneal, 1.0,
                           0.75, 1.0, float(np.float32(0.7)), rng_step=2 * step, d_rng_step=2 * step + 1).cpu().numpy()
        assert (float(curr_g_loss), float(t1), float(t2)) == tuple(float(np.float32(x)) for x in want[:3])
        assert abs(curr_g_loss - (t1 + t2)) < 1e-5 * abs(curr_g_loss) and t2 < 0
    for a, b in zip(eng.g_p + eng.d_p, twin.g_p + twin.d_p):
        assert torch.equal(a, b)
    assert eng.adam_t == twin.adam_t == 2 * NB                           # one shared Adam step counter (Q5)

    # validation (train.py:333-346)
    X_vad = vad_data_tr[:200].toarray().astype("float32")
    pred_vad = sess.run(generator_out, feed_dict={generator_network.input_ph: X_vad})
    pred_vad[X_vad.nonzero()] = -np.inf
    ndcg = O.ndcg_binary_at_k(pred_vad.astype(np.float64), vad_data_te[:200].toarray(), 100)
    assert np.isfinite(ndcg).all() and 0.0 <= np.nanmean(ndcg) <= 1.0

    with pytest.raises(NotImplementedError):
        sess.run([y_data], feed_dict={})


def test_factories_with_reference_argument_lists_and_non_default_config(tmp_path, monkeypatch):
    """train.py:130,136 / test.py:79,85: `generator(pro_dir)` + `discriminator(n_items, FEATURE_LEN, h0, h1, h2, h3)` with
    nothing else passed.  A non-default ./config.ini is picked up by generator(); sizes that differ from it re-create the
    discriminator part; a D step then runs with those sizes."""
    import torch
    from ltgan.dataset import materialize_askubuntu
    from ltgan.discriminator import discriminator
    from ltgan.engine import Pairs
    from ltgan.generator import current_engine, generator, reset_default_graph

    pro_dir = str(tmp_path / "Askubuntu_Sample") + "/"
    materialize_askubuntu(os.path.join(ROOT, "tests", "golden", "askubuntu_raw.npz"), pro_dir)
    (tmp_path / "config.ini").write_text("[Long-Tail-GAN]\nh0_size = 64\nh1_size = 96\nh2_size = 160\nh3_size = 128\nNUM_EPOCH = 8\n"
                                         "BATCH_SIZE = 100\nDISPLAY_ITER = 10\nLEARNING_RATE = 0.0003\nto_restore = 0\nmodel_name = LTG\nGANLAMBDA = 1.0\n")
    monkeypatch.chdir(tmp_path)
    reset_default_graph()
    with pytest.raises(ValueError):
        discriminator(1000, 1000, 64, 96, 160, 128)          # no graph yet
    net, probs, loss, g_params, p_dims, tas, cap = generator(pro_dir)
    eng = current_engine()
    assert eng is net.engine and (eng.h0, eng.h1, eng.h2, eng.h3) == (64, 96, 160, 128) and abs(eng.cfg.lr - 3e-4) < 1e-9
    assert p_dims == [200, 600, 1000] and tas == 20000 and cap == 0.2
    out = discriminator(1000, 1000, 64, 96, 160, 128)        # same sizes: the engine's discriminator is used as is
    assert len(out) == 9 and [tuple(p.shape) for p in out[2]] == [(64, 96), (96,), (64, 160), (160,), (256, 128), (128,), (128, 1), (1,)]
    w_before = eng.d_p[0].clone()
    out = discriminator(1000, 1000, 32, 48, 80, 64)          # other sizes: re-created
    assert (eng.h0, eng.h1, eng.h2, eng.h3) == (32, 48, 80, 64) and eng.cfg.d_h3 == 64 and tuple(eng.d_emb.shape) == (1000, 32)
    assert [tuple(p.shape) for p in out[2]] == [(32, 48), (48,), (32, 80), (80,), (128, 64), (64,), (64, 1), (1,)]
    assert tuple(w_before.shape) == (64, 96)
    rng = np.random.default_rng(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(eng.device)
    real = Pairs(t(rng.integers(0, 1000, 300)), t(rng.integers(0, 1000, 300)))
    fake = Pairs(t(rng.integers(0, 1000, 280)), t(rng.integers(0, 1000, 280)))
    l = float(eng.d_step(real, fake, 0.7, rng_step=1)[0].item())
    assert np.isfinite(l) and abs(l - 580 * np.log(2.0)) < 0.25 * 580 * np.log(2.0)      # y ~ 0.5 at initialisation
    with pytest.raises(ValueError):
        discriminator(999, 1000, 32, 48, 80, 64)             # another item count than the generator's
