"""Multi-epoch parity: the device Trainer on Askubuntu_Sample (rebuilt from the committed fixture)
against the fp64 oracle trajectory tests/golden/oracle_trajectory.npz (oracle/trajectory.py, made
by tests/golden/make_trajectory.py).  Same initial weights, same counter-RNG streams, same batch
order.  Bounds: loss triplet 1e-3 relative per sub-epoch, NDCG@100 within 0.002."""
import os

import numpy as np
import pytest

from oracle import ltg_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_two_epoch_trajectory_matches_oracle(tmp_path):
    import torch
    import helpers as Hh
    from ltgan import data_processing as dp
    from ltgan.dataset import DeviceData, EvalData, IndexData, materialize_askubuntu
    from ltgan.engine import Engine
    from ltgan.trainer import Evaluator, Trainer
    gold = np.load(os.path.join(G, "oracle_trajectory.npz"))
    d = str(tmp_path / "ds")
    materialize_askubuntu(os.path.join(G, "askubuntu_raw.npz"), d)
    idx = IndexData.from_dir(d)
    hs = tuple(int(x) for x in gold["hs"])
    eng = Engine(idx.n_items, h_sizes=hs, lr=float(gold["lr"]), precision="bf16", seed=int(gold["seed"]))
    P = O.init_generator(idx.n_items, seed=int(gold["gen_seed"]))
    D = O.init_discriminator(idx.n_items, *hs, seed=int(gold["disc_seed"]))
    eng.set_generator(Hh.gen_to_engine(P))
    emb, darr = Hh.disc_to_engine(D)
    eng.set_discriminator(emb, darr)
    data = DeviceData(idx, 100, eng.device)
    S = int(gold["S"])
    tr = Trainer(eng, data, num_sub_epochs=S, shuffle_seed=0)
    vtr, vte, _ = dp.load_tr_te_data(os.path.join(d, "validation_tr.csv"), os.path.join(d, "validation_te.csv"), idx.n_items)
    ev = Evaluator(eng, EvalData(vtr, vte, eng.device))
    for e in range(int(gold["epochs"])):
        tr.create_phase()
        cnt = data.fake_cnt.cpu().numpy()
        want_cnt = gold["e%d_cnt" % e]
        assert np.abs(cnt - want_cnt).sum() <= 2, (e, np.abs(cnt - want_cnt).sum())      # sampler flips are rare
        gen = data.fake_gen.cpu().numpy()
        same = np.array_equal(np.sort(gen[gen >= 0]), np.sort(gold["e%d_fake_gen" % e].astype(np.int64)))
        assert np.array_equal(tr.order, gold["e%d_order" % e])
        dl = tr.d_phase().cpu().numpy()[:S, 0]
        gl = tr.g_phase().cpu().numpy()[:S, :3]
        tol = 1e-3 if same else 3e-3
        np.testing.assert_allclose(dl, gold["e%d_d_loss" % e], rtol=tol)
        want_g = gold["e%d_g_loss" % e]
        np.testing.assert_allclose(gl[:, 0], want_g[:, 0], rtol=tol)
        np.testing.assert_allclose(gl[:, 1], want_g[:, 1], rtol=tol)
        np.testing.assert_allclose(gl[:, 2], want_g[:, 2], rtol=5 * tol, atol=1e-5)     # small-magnitude GAN term
        assert abs(tr.last_anneal[-1] - want_g[-1, 3]) < 1e-12
        m = ev.run(rng_step=1000 + e)
        wm = gold["e%d_metrics" % e]
        assert abs(m["ndcg"] - wm[0]) < 2e-3 and abs(m["recall20"] - wm[1]) < 2e-3 and abs(m["recall50"] - wm[2]) < 2e-3, (m, wm)
    np.testing.assert_allclose(eng.g_p[7].cpu().numpy()[:64], gold["final_bp1_head"], atol=2e-5)
    # 404 Adam steps of lr 1e-4 travel up to 4e-2; fp32-vs-fp64 drift through the sign-like Adam normalisation stays ~1% of that
    np.testing.assert_allclose(eng.d_p[6].cpu().numpy()[:64], gold["final_w4_head"], atol=5e-4)


def test_first_fifty_steps_of_each_phase_match_oracle(tmp_path):
    """SURVEY 8/d6 literally: the loss of EACH of the first 50 discriminator updates and the loss triplet of EACH of the first 50 generator
    updates of global epoch 0 under config.ini's schedule (S = 10: the 50 G steps sit behind all 1 010 D steps; train.py:287-329) against
    tests/golden/oracle_trajectory_steps.npz (tests/golden/make_trajectory_steps.py) at 1e-3 relative -- a drift that cancels by the sub-epoch's
    last step (what the two-epoch test compares) would show here."""
    import torch
    import helpers as Hh
    from ltgan.dataset import DeviceData, IndexData, materialize_askubuntu
    from ltgan.engine import Engine
    from ltgan.trainer import Trainer
    gold = np.load(os.path.join(G, "oracle_trajectory_steps.npz"))
    d = str(tmp_path / "ds")
    materialize_askubuntu(os.path.join(G, "askubuntu_raw.npz"), d)
    idx = IndexData.from_dir(d)
    hs = tuple(int(x) for x in gold["hs"])
    n = int(gold["n_steps"])
    eng = Engine(idx.n_items, h_sizes=hs, lr=float(gold["lr"]), precision="bf16", seed=int(gold["seed"]))
    eng.set_generator(Hh.gen_to_engine(O.init_generator(idx.n_items, seed=int(gold["gen_seed"]))))
    emb, darr = Hh.disc_to_engine(O.init_discriminator(idx.n_items, *hs, seed=int(gold["disc_seed"])))
    eng.set_discriminator(emb, darr)
    data = DeviceData(idx, 100, eng.device)
    tr = Trainer(eng, data, num_sub_epochs=int(gold["S"]), shuffle_seed=0, step_log=n)
    tr.create_phase()
    cnt = data.fake_cnt.cpu().numpy()
    assert np.abs(cnt - gold["cnt"]).sum() <= 2                      # sampler flips are rare
    gen = data.fake_gen.cpu().numpy()
    same = np.array_equal(np.sort(gen[gen >= 0]), np.sort(gold["fake_gen"].astype(np.int64)))
    assert np.array_equal(tr.order, gold["order"])
    tol = 1e-3 if same else 3e-3
    sub = tr.d_phase().cpu().numpy()[:, 0]
    dl = tr.d_step_log.cpu().numpy()[:n, 0]
    assert len(tr.order) * tr.S == int(gold["d_steps_total"])
    err_d = np.max(np.abs(dl - gold["d_loss_steps"]) / np.abs(gold["d_loss_steps"]))
    np.testing.assert_allclose(dl, gold["d_loss_steps"], rtol=tol)
    np.testing.assert_allclose(sub, gold["d_loss_sub_epochs"], rtol=tol)       # every sub-epoch's last step too
    tr.g_phase()
    gl = tr.g_step_log.cpu().numpy()[:n, :3]
    want = gold["g_loss_steps"]
    err_g = np.max(np.abs(gl[:, :2] - want[:, :2]) / np.abs(want[:, :2]))
    print("first %d steps, same fake pairs %s: worst relative error d_loss %.2e, g / vae loss %.2e, gan loss %.2e" %
          (n, same, err_d, err_g, np.max(np.abs(gl[:, 2] - want[:, 2]) / np.maximum(np.abs(want[:, 2]), 1e-5))))
    np.testing.assert_allclose(gl[:, 0], want[:, 0], rtol=tol)
    np.testing.assert_allclose(gl[:, 1], want[:, 1], rtol=tol)
    np.testing.assert_allclose(gl[:, 2], want[:, 2], rtol=5 * tol, atol=1e-5)  # small-magnitude GAN term (as in the two-epoch test)


@pytest.mark.parametrize("workload,users,hs", [("custom:1000", 450, (20, 24, 40, 36)), ("custom:8264", 230, (20, 24, 40, 36)),
                                               ("custom:1000", 450, (100, 150, 250, 300))])     # config.ini's sizes: the LDS-staged tower kernels
def test_hoisted_phase_work_is_bit_identical_to_the_step_by_step_loop(workload, users, hs):
    """Two reorderings of work that no weight update separates: phase C over spans of batches (one forward + one sampler launch,
    every batch with its own RNG counter and row numbers) and every fake tower of phase G evaluated ahead (the discriminator is
    fixed during the phase; ltg_fake_tower_batched + ltg_g_opts.y_pre).  Two epochs with both against two epochs batch by batch
    and tower-in-step: identical fake pairs, losses and parameters, bit for bit (small item slab: nine-launch G step; 8 264
    items: streaming decoder path + lazy Adam clock)."""
    import torch
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    idx, _ = synthetic_index(workload, users=users, seed=9)
    runs = []
    for hoist in (False, True):
        eng = Engine(idx.n_items, h_sizes=hs, lr=1e-3, seed=5, d_seed=2)
        tr = Trainer(eng, DeviceData(idx, 100, eng.device), num_sub_epochs=2, shuffle_seed=3, span_create=hoist, batched_tower=hoist)
        assert tr.batched_tower == hoist and (tr.span_batches > 1) == hoist
        out = []
        for _ in range(2):
            tr.create_phase()
            out += [tr.data.fake_gen.clone(), tr.data.fake_pop.clone(), tr.data.fake_cnt.clone()]
            out.append(tr.d_phase().clone())
            out.append(tr.g_phase().clone())
        torch.cuda.synchronize()
        runs.append([x.cpu() for x in out + eng.g_p + eng.d_p + eng.g_m])
    for k, (a, b) in enumerate(zip(*runs)):
        assert torch.equal(a, b), k


def test_pipelined_g_phase_equals_the_step_by_step_loop(monkeypatch):
    """Large item slabs run phase G (train.py:307-329) as one ltg_g_step_sharded call per step -- decoder weight update and lazy-clock slice
    forked beside the NEXT step, the next batch's rows of W_q0 caught up AHEAD on the side stream (no catch-up launch of its own; also with
    LTGAN_Q0_AHEAD=0: every call catches up its own rows) -- with every fake tower evaluated ahead.  Two global epochs (C, S x D, S x G) of
    the real loop on a synthetic 9 000-item dataset: identical fake pairs, and every generator tensor, Adam moment and loss equals the loop
    over ltg_g_step bit for bit."""
    import torch
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    idx, _ = synthetic_index("custom:9000", users=430, seed=21)
    runs = []
    for pipe in (False, True, "own-catch-up"):
        monkeypatch.setenv("LTGAN_Q0_AHEAD", "0" if pipe == "own-catch-up" else "1")
        eng = Engine(idx.n_items, h_sizes=(20, 24, 40, 36), lr=1e-3, precision="bf16", seed=5, d_seed=9)
        data = DeviceData(idx, 100, eng.device)
        tr = Trainer(eng, data, num_sub_epochs=3, shuffle_seed=4, pipe_step=bool(pipe))
        assert (tr.pipe is not None) == bool(pipe) and eng.lazy_q0
        losses = []
        for _ in range(2):
            tr.create_phase()
            losses.append(tr.d_phase().clone())
            losses.append(tr.g_phase().clone())
        torch.cuda.synchronize()
        tr.check_pipe()                                                    # no device-side wait of the hand-overs gave up
        if pipe:
            assert tr.pipe.handover in ("device-words + tail stream", "device-words", "events") and tr.pipe.expired_waits() == 0, tr.pipe.handover
            steps = int(tr.update_count)          # G steps so far (two phases)
            if pipe is True and tr.pipe.handover != "events":
                # every step but the first of each phase found its rows caught up by the step before
                assert tr.pipe.ahead_calls == steps - 2, (tr.pipe.ahead_calls, steps)
            else:
                assert tr.pipe.ahead_calls == 0
        runs.append((data.fake_gen.clone(), data.fake_pop.clone(), losses, [t.clone() for t in eng.g_p + eng.g_m + eng.g_v + eng.d_p]))
    a = runs[0]
    for b in runs[1:]:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for x, y in zip(a[2], b[2]):
            assert torch.equal(x, y)
        for k, (x, y) in enumerate(zip(a[3], b[3])):
            assert torch.equal(x, y), k


def test_small_slab_loop_with_the_forked_discriminator_step_is_bit_identical():
    """Small item slabs (Askubuntu_Sample's regime): ltg_d_step runs its backward's jobs B / C on the aux stream (Engine.d_fork).  Two global
    epochs of the real loop with and without the fork: identical fake pairs, losses and every tensor of both models."""
    import torch
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    idx, _ = synthetic_index("custom:1000", users=430, seed=23)
    runs = []
    for fork in (False, True):
        eng = Engine(idx.n_items, lr=1e-3, precision="bf16", seed=5, d_seed=9)
        eng.d_fork = fork
        data = DeviceData(idx, 100, eng.device)
        tr = Trainer(eng, data, num_sub_epochs=3, shuffle_seed=4)
        assert tr.pipe is None and tr.batched_tower
        losses = []
        for _ in range(2):
            tr.create_phase()
            losses.append(tr.d_phase().clone())
            losses.append(tr.g_phase().clone())
        torch.cuda.synchronize()
        if fork:
            assert eng._dfork is not None
            if not eng._dfork.ok:
                pytest.skip("the aux stream shares a hardware queue with the caller's stream on this box")
            assert eng._dfork.expired_waits() == 0 and eng._dfork.seq > 0
        else:
            assert eng._dfork is None
        runs.append((data.fake_gen.clone(), losses, [t.clone() for t in eng.g_p + eng.g_m + eng.g_v + eng.d_p + eng.d_m + eng.d_v]))
    a, b = runs
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    for k, (x, y) in enumerate(zip(a[2], b[2])):
        assert torch.equal(x, y), k


def test_an_expired_device_side_wait_is_reported_by_the_trainer():
    """The hand-overs of the one-call G step poll words of device memory with a bound; a poll that gives up is counted in
    ltg_pipe.sync[2] and must surface as an error at the end of the phase (Trainer.check_pipe), not as silently wrong weights."""
    import torch
    from ltgan.dataset import DeviceData
    from ltgan.engine import Engine
    from ltgan.synthetic import synthetic_index
    from ltgan.trainer import Trainer
    idx, _ = synthetic_index("custom:9000", users=230, seed=22)
    eng = Engine(idx.n_items, h_sizes=(20, 24, 40, 36), lr=1e-3, precision="bf16", seed=5, d_seed=9)
    data = DeviceData(idx, 100, eng.device)
    tr = Trainer(eng, data, num_sub_epochs=2, shuffle_seed=4, pipe_step=True)
    assert tr.pipe is not None
    tr.epoch()                                   # a clean epoch passes the check
    assert tr.pipe.handover in ("device-words + tail stream", "device-words", "events")
    if not tr.pipe.handover.startswith("device-words"):
        tr.pipe.sync[2] = 3
        with pytest.raises(RuntimeError, match="gave up"):
            tr.epoch()
        return
    torch.cuda.synchronize()
    before = [t.clone() for t in eng.g_p + eng.g_m + eng.g_v]
    tr.pipe.sync[2] = 3                          # as if three polls had given up: the pipe is poisoned
    with pytest.raises(RuntimeError, match="gave up"):
        tr.epoch()
    torch.cuda.synchronize()
    # ... and every kernel of the poisoned phase that writes the generator returned at once: the model is the one from before
    for k, (x, y) in enumerate(zip(before, eng.g_p + eng.g_m + eng.g_v)):
        assert torch.equal(x, y), ("generator tensor moved behind a wait that gave up", k)
    # nothing is read out either: the checkpoint writer and the parameter accessor check the pipes first
    from ltgan.train import save_checkpoint
    with pytest.raises(RuntimeError, match="gave up"):
        save_checkpoint("/tmp/ltgan_never_written.pt", eng, tr, 0)
    assert not os.path.exists("/tmp/ltgan_never_written.pt")
    # Pipe.reset (what Engine.g_step_sharded does after a call that returned an error): nothing in flight, every word zero, ordinals restart
    tr.pipe.reset()
    assert tr.pipe.expired_waits() == 0 and tr.pipe.c.seq == 0 and int(tr.pipe.sync.abs().sum().item()) == 0


@pytest.mark.parametrize("queues,flags,expect", [("1", "0", "events\n"), ("2", "0", "device-words\n"), ("2", "128", "device-words\n"),
                                                 ("4", "128", "device-words + tail stream\n")])
def test_one_call_step_falls_back_to_events_when_its_streams_share_a_hardware_queue(queues, flags, expect):
    """HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues.  With ONE, a waiter on the side stream sits in front of its producer in
    the same queue: ltg_g_pipe_probe must see that and the engine must fall back to event pairs; with TWO the side stream gets a queue
    of its own but the Adam tail's stream (LTG_PIPE_TAIL_OWN = 128, opt-in) cannot (the tail then stays on the caller's stream); with four
    all three streams run concurrently -- with the same bits as the step-by-step loop in every mode (scripts/soak_onecall.py compares every tensor).  A fresh
    process: the variable is read at HIP start-up."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, GPU_MAX_HW_QUEUES=queues, LTGAN_PIPE_FLAGS=flags)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_onecall.py"), "9000", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.replace(" expired", "\n expired")
    # (the probe also tests the side stream against the tail stream: in one queue the Adam tail would sit behind the weight update)
    assert "hand-over: " + expect in out and "expired waits: 0" in out and "differing: []" in out, r.stdout
