"""Item-sharded multi-GPU path on the GPU box: 2 ranks (gloo backend, both on cuda:0 -- a single-GPU box
cannot host two RCCL ranks) against the unsharded trainer.  See tests/dist_shard_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,users,precision,mode", [("custom:1000", 250, "fp32", ""), ("ml20m", 200, "fp32", ""), ("ml20m", 200, "bf16", ""),
                                                           ("custom:1000", 250, "fp32", "dsplit"), ("ml20m", 200, "bf16", "wide_fp8"),
                                                           ("ml20m", 200, "bf16", "cutpoints"), ("c4", 200, "bf16", ""),
                                                           ("ml20m", 200, "bf16", "wide_fp8_full"), ("ml20m", 200, "bf16", "tail_own")])
def test_two_rank_item_sharding_matches_unsharded(workload, users, precision, mode):
    """mode "dsplit": the discriminator's pair rows are split over the two ranks too (ltg_d_grad -> gradient all-reduce ->
    ltg_d_apply) -- same d_loss trajectory and weights as the unsharded step.  mode "wide_fp8": BASELINE config 5's shape inside
    the sharded loop -- a wide discriminator with fp8 GEMM operands (operand-format shadows maintained by the Adam sweep), pair-split,
    over item slabs large enough for the streaming decoder kernels and the lazy Adam clock of W_q0; "wide_fp8_full": the same with
    config 5's full sizes (2048, 1024, 512, 256).  bf16 slabs of 8192 items or more run the G step as ONE call with the exchanges
    issued from inside it (ltg_g_step_sharded; here through host callbacks over gloo); "cutpoints": the five-call sequence with
    torch.distributed collectives instead.  ("c4", 200 users): BASELINE config 4's 200 000 items, 100 000 per rank.  "tail_own": the Adam tail on the pipe's third stream
    (LTG_PIPE_TAIL_OWN; the default keeps it on the caller's stream)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "tail_own":
        env["LTGAN_PIPE_FLAGS"] = "128"
        mode = ""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "tests", "dist_shard_worker.py"), workload, str(users), precision] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "SHARDED_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


@pytest.mark.parametrize("workload,users,precision,mode", [("c4", 200, "bf16", ""), ("ml20m", 200, "bf16", ""), ("c4", 200, "bf16", "wide_fp8_full"),
                                                           ("custom:65600", 200, "bf16", "")])
def test_eight_rank_item_sharding_matches_unsharded(workload, users, precision, mode):
    """World size 8 -- the node the split is built for -- on the rig a 1-GPU box allows: eight gloo ranks sharing cuda:0, every rank also
    running the unsharded trainer it is compared with.  c4: 200 000 items = seven slabs of 25 024 and one of 24 832 (the uneven last
    slab), `rowpart_all [8][B][5]`, the one-call step on every rank; with "wide_fp8_full" config 5's discriminator pair-split over 8
    ranks.  ml20m: 20 000 items = 2 560-item slabs, below the one-call step's 8 192 -- every rank agrees on the cut-point sequence
    (the worker asserts which path ran).  custom:65600: slabs of 8 256 items and a last one of 7 808 < 8 192 -- ONE rank's slab is too
    small for the one-call step, so all eight must agree on the cut-point sequence (`ShardedTrainer._init_sharded_step`)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(ROOT, "tests", "dist_shard_worker.py"), workload, str(users), precision] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=2400)      # fresh children only
    assert out.returncode == 0 and "SHARDED_OK world=8" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


@pytest.mark.parametrize("world,workload,users", [(2, "c4", 200), (8, "c4", 200)])
def test_one_shot_exchange_equals_the_rank_ordered_host_transport(world, workload, users):
    """The second transport behind ltg_comm (csrc/ltg_oneshot.h, ltgan._rccl.OneShotComm; SURVEY 8/e1's one-shot exchange for the <= 240-KB
    messages): every rank's staging buffer mapped into every peer through HIP IPC, one kernel per exchange, contributions added in rank order.
    Against host callbacks that add in the same order the two-epoch run must leave the same bits on every rank -- fake pairs, losses, every
    generator tensor and Adam moment -- at world sizes 2 and 8 (c4: 100 000-item slabs / seven slabs of 25 024 items and one of 24 832).
    Correctness only: RCCL stays the transport of every measurement."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", LTGAN_ONESHOT_LIMIT_MS="20000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29583", os.path.join(ROOT, "tests", "dist_oneshot_worker.py"), workload, str(users)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)      # fresh children only
    assert out.returncode == 0 and ("ONESHOT_OK world=%d" % world) in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


def test_one_shot_exchange_gives_up_and_raises_when_a_peer_never_sends():
    """The one-shot transport's device-side waits are bounded (ltg_oneshot.limit_ms): with a peer that sits out the G phase every exchange of the
    other rank gives up after its bound, counts it in the stage, and the phase ends in an LtgError (ShardedTrainer.g_phase) -- a forced expiry,
    not a hang."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", LTGAN_ONESHOT_LIMIT_MS="40")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29585", os.path.join(ROOT, "tests", "dist_oneshot_worker.py"), "c4", "200", "expiry"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "ONESHOT_EXPIRY_OK world=2" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


def test_direct_rccl_transport_of_the_one_call_step_at_world_size_one():
    """The transport a GPU node uses: RCCL bound directly, its entry points called by the library in-stream (tests/dist_rccl_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_rccl_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "RCCL_DIRECT_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


@pytest.mark.parametrize("workload,users,mode", [("custom:50048", 200, ""), ("c4", 200, ""), ("custom:50048", 200, "wide_fp8")])
def test_two_ranks_on_two_gpus_over_rccl(workload, users, mode):
    """The same comparison on a box with at least two GPUs: backend nccl (= RCCL), one GPU per rank, the three exchanges of every
    G step issued in-stream by RCCL's own entry points (ltgan._rccl.RcclComm; the worker asserts ncclCommCount == 2 and that no
    device-side wait of the hand-overs gave up).  custom:50048 = two slabs of 25 024 items (what one of eight ranks owns at 200 000);
    c4 = 100 000 items per rank.  Skipped on single-GPU boxes: RCCL refuses two ranks on one device."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", LTGAN_TEST_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "tests", "dist_shard_worker.py"), workload, str(users), "bf16"] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)      # fresh children: this process never hands its GPU state on
    assert out.returncode == 0 and "SHARDED_OK" in out.stdout and "transport=rccl-direct" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]
