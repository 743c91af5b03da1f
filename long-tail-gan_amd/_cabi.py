"""ctypes binding of the C ABI declared in include/ltg.h (libltg_hip.so).

There is deliberately NO fallback: if the HIP library is missing or a symbol is absent the import
of the product path fails loudly (the oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTG_HIP_LIB") or os.path.join(_HERE, "libltg_hip.so")   # LTG_HIP_LIB: another build of the SAME library (A/B timing)

LTG_PREC_BF16 = 0
LTG_PREC_FP32 = 1
LTG_PREC_FP8 = 2
LTG_DARITH_FP32, LTG_DARITH_BF16X6, LTG_DARITH_BF16X4 = 0, 1, 2    # ltg_config.d_arith
LTG_ABI_VERSION = 14
LTG_PLAN_AHEAD, LTG_PLAN_SHADOW = 1, 2
LTG_Q0_HIST = 1024

ERRORS = {0: "LTG_OK", -1: "LTG_EINVAL", -2: "LTG_EWORKSPACE", -3: "LTG_ELAUNCH"}

vp = C.c_void_p


class ltg_config(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("h_enc", C.c_int32), ("z_dim", C.c_int32), ("d_feat", C.c_int32),
                ("d_h0", C.c_int32), ("d_h1", C.c_int32), ("d_h2", C.c_int32), ("d_h3", C.c_int32),
                ("precision", C.c_int32), ("tuning", C.c_int32), ("item_lo", C.c_int32), ("n_items_global", C.c_int32),
                ("d_precision", C.c_int32), ("d_arith", C.c_int32),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("seed", C.c_uint64)]


class ltg_gen_state(C.Structure):
    _fields_ = [("p", vp * 8), ("m", vp * 8), ("v", vp * 8), ("wp1t_bf16", vp),
                ("q0_last", vp), ("q0_lr_hist", vp), ("q0_ord", C.c_int32), ("q0_period", C.c_int32)]   # optional lazy Adam clock of W_q0


class ltg_disc_state(C.Structure):
    _fields_ = [("emb", vp), ("p", vp * 8), ("m", vp * 8), ("v", vp * 8),
                ("emb_fp8", vp), ("w1t_fp8", vp), ("w2t_fp8", vp), ("w3t_fp8", vp), ("w3_fp8", vp)]      # optional e4m3 operand shadows


class ltg_batch(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_unique", C.c_int32), ("indptr", vp), ("indices", vp), ("values", vp),
                ("slot", vp), ("uptr", vp), ("rowidx", vp), ("csr_pos", vp), ("row_norm2", vp), ("uitem", vp)]


class ltg_gen_acts(C.Structure):
    _fields_ = [("h1", vp), ("mulv", vp), ("z", vp), ("h2", vp), ("logits", vp), ("lse", vp), ("kl_rows", vp),
                ("row_scale", vp)]


class ltg_probe(C.Structure):
    _fields_ = [("kernel_id", C.c_int32), ("reserved0", C.c_int32), ("ev_start", vp), ("ev_stop", vp)]


KERNEL_IDS = {"enc0_fwd": 1, "enc1": 2, "dec0": 3, "dec1_fwd": 4, "d_l1": 5, "d_l2": 6, "d_bwd1": 7, "d_bwd2": 8, "d_adam": 9,
              "dh2": 10, "dec1_bwd_adam": 11, "enc0_bwd_adam": 12, "dz": 13, "dh1": 14, "wgrad_p0": 15, "wgrad_q1": 16,
              "row_dlogits": 17, "enc0_grad": 18, "g_tail": 19, "exch_h1": 20, "exch_rowpart": 21, "exch_dh2": 22}


class ltg_fwd_opts(C.Structure):
    _fields_ = [("keep_prob", C.c_float), ("is_training", C.c_float), ("rng_step", C.c_uint64), ("drop_keep", vp),
                ("eps", vp), ("probe", C.POINTER(ltg_probe)), ("rows_per_step", C.c_int32), ("reserved0", C.c_int32)]


class ltg_pairs(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved0", C.c_int32), ("pop", vp), ("niche", vp), ("row", vp)]


class ltg_d_opts(C.Structure):
    _fields_ = [("keep_prob", C.c_float), ("adam_t", C.c_int32), ("rng_step", C.c_uint64), ("drop_real", vp * 3),
                ("drop_fake", vp * 3), ("probe", C.POINTER(ltg_probe)), ("aux_stream", vp), ("sync", vp), ("seq", C.c_uint32), ("reserved0", C.c_int32)]


class ltg_g_opts(C.Structure):
    _fields_ = [("fwd", ltg_fwd_opts), ("anneal", C.c_float), ("gan_lambda", C.c_float), ("d_keep_prob", C.c_float),
                ("adam_t", C.c_int32), ("d_rng_step", C.c_uint64), ("drop_fake", vp * 3), ("cnt", vp), ("probe", C.POINTER(ltg_probe)),
                ("aux_stream", vp), ("ev_fork", vp), ("ev_join", vp), ("dec1_done", C.c_int32), ("fake_done", C.c_int32), ("ev_sweep", vp), ("y_pre", vp)]


class ltg_sample_inputs(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("max_cand", C.c_int32), ("cand_ptr", vp), ("cand_idx", vp), ("pop_ptr", vp),
                ("pop_idx", vp), ("n_sample", vp), ("slot_ptr", vp), ("valid_item", vp), ("rng_step", C.c_uint64),
                ("u_gumbel", vp), ("u_pick", vp), ("cand_logit", vp), ("rows_per_step", C.c_int32), ("reserved0", C.c_int32)]


# the two collective entry points a caller hands to ltg_g_step_sharded: the signatures of ncclAllReduce / ncclAllGather (rccl.h:611, :678)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_size_t, C.c_int, vp, vp)
LTG_NCCL_FLOAT32, LTG_NCCL_SUM = 7, 0
LTG_PIPE_NO_DEC1_FORK, LTG_PIPE_NO_SLICE_FORK, LTG_PIPE_WIDE_GRAD, LTG_PIPE_EVENTS, LTG_PIPE_SLICE_IN_TOUCH, LTG_PIPE_TAIL_OWN = 1, 2, 8, 16, 32, 128


class ltg_comm(C.Structure):
    _fields_ = [("comm", vp), ("n_ranks", C.c_int32), ("rank", C.c_int32), ("all_reduce", vp), ("all_gather", vp)]


class ltg_pipe(C.Structure):
    _fields_ = [("side_stream", vp), ("ev_fork", vp), ("ev_dec1", vp), ("ev_tail", vp),
                ("h1pre", vp), ("rowpart_all", vp), ("dh2", vp), ("flags", C.c_int32), ("seq", C.c_uint32), ("sync", vp), ("tail_stream", vp),
                ("q0_mark", vp), ("next_uitem", vp), ("next_nu", C.c_int32), ("caught_up", C.c_int32), ("shadow_out", vp)]


LTG_ONESHOT_MAX_RANKS = 16


class ltg_oneshot(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("rank", C.c_int32), ("seq", C.c_uint32), ("limit_ms", C.c_uint32), ("max_floats", C.c_size_t),
                ("stage", vp * LTG_ONESHOT_MAX_RANKS)]


# every symbol include/ltg.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ltg_oneshot_stage_bytes": (C.c_size_t, [C.c_int32, C.c_size_t]),
    "ltg_oneshot_expired_offset": (C.c_size_t, [C.c_int32]),
    "ltg_oneshot_all_reduce": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]),
    "ltg_oneshot_all_gather": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp, vp]),
    "ltg_g_step_sharded_ok": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.c_int32]),
    "ltg_g_step_sharded_plan": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch), C.POINTER(ltg_pipe)]),
    "ltg_g_step_sharded": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_disc_state), C.POINTER(ltg_batch),
                                     C.POINTER(ltg_pairs), C.POINTER(ltg_g_opts), C.POINTER(ltg_gen_acts), C.POINTER(ltg_comm),
                                     C.POINTER(ltg_pipe), vp, vp, C.c_size_t, vp]),
    "ltg_g_pipe_join": (C.c_int, [C.POINTER(ltg_pipe), vp]),
    "ltg_g_pipe_probe": (C.c_int, [C.POINTER(ltg_pipe), vp]),
    "ltg_abi_version": (C.c_int32, []),
    "ltg_workspace_bytes": (C.c_size_t, [C.POINTER(ltg_config), C.c_int32, C.c_int32]),
    "ltg_vae_forward": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch),
                                  C.POINTER(ltg_fwd_opts), C.POINTER(ltg_gen_acts), vp, vp, C.c_size_t, vp]),
    "ltg_sample_pairs": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_sample_inputs), vp, vp, vp, vp, vp, vp]),
    "ltg_d_step": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), C.POINTER(ltg_pairs),
                             C.POINTER(ltg_pairs), C.POINTER(ltg_d_opts), vp, vp, C.c_size_t, vp]),
    "ltg_d_grad_floats": (C.c_size_t, [C.POINTER(ltg_config)]),
    "ltg_d_grad": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), C.POINTER(ltg_pairs), C.POINTER(ltg_pairs), C.c_int32,
                             C.c_int32, C.POINTER(ltg_d_opts), vp, vp, C.c_size_t, vp]),
    "ltg_d_apply": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), vp, C.c_int32, vp, vp]),
    "ltg_g_bwd_dec1": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch), C.POINTER(ltg_pairs),
                                 C.POINTER(ltg_g_opts), C.POINTER(ltg_gen_acts), vp, C.c_size_t, vp]),
    "ltg_g_step": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_disc_state),
                             C.POINTER(ltg_batch), C.POINTER(ltg_pairs), C.POINTER(ltg_g_opts),
                             C.POINTER(ltg_gen_acts), vp, vp, C.c_size_t, vp]),
    "ltg_g_fwd_enc": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch), C.POINTER(ltg_fwd_opts),
                                C.POINTER(ltg_gen_acts), vp]),
    "ltg_g_fwd_rest": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch), C.POINTER(ltg_pairs),
                                 C.POINTER(ltg_fwd_opts), C.POINTER(ltg_gen_acts), vp, vp, C.c_size_t, vp]),
    "ltg_rowstats_combine": (C.c_int, [C.POINTER(ltg_config), vp, C.c_int32, C.c_int32, vp, vp, C.c_size_t, vp]),
    "ltg_g_bwd_dec": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_disc_state), C.POINTER(ltg_batch),
                                C.POINTER(ltg_pairs), C.POINTER(ltg_g_opts), C.POINTER(ltg_gen_acts), vp, C.c_int32, vp, vp, vp,
                                C.c_size_t, vp]),
    "ltg_g_bwd_rest": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), C.POINTER(ltg_batch), C.POINTER(ltg_pairs),
                                 C.POINTER(ltg_g_opts), C.POINTER(ltg_gen_acts), vp, vp, C.c_size_t, vp]),
    "ltg_gather_cand_logits": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_sample_inputs), vp, vp, vp]),
    "ltg_refresh_shadow": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), vp]),
    "ltg_forward_scratch_bytes": (C.c_size_t, [C.POINTER(ltg_config), C.c_int32]),
    "ltg_g_fake_tower": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), C.POINTER(ltg_pairs), C.POINTER(ltg_g_opts), C.c_int32, vp, C.c_size_t, vp]),
    "ltg_fake_tower_batched": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), C.POINTER(ltg_pairs), vp, vp, vp, C.c_float, vp, vp,
                                        C.c_size_t, vp]),
    "ltg_g_flush": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_gen_state), vp]),
    "ltg_refresh_d_shadow": (C.c_int, [C.POINTER(ltg_config), C.POINTER(ltg_disc_state), vp]),
    "ltg_rank_metrics": (C.c_int, [C.POINTER(ltg_config), vp, C.POINTER(ltg_batch), C.POINTER(ltg_batch), C.c_int32,
                                   C.c_int32, C.c_int32, vp, vp]),
    "ltg_rank_scores": (C.c_int, [C.POINTER(ltg_config), vp, C.POINTER(ltg_batch), C.POINTER(ltg_batch), vp, vp]),
    "ltg_rank_counts": (C.c_int, [C.POINTER(ltg_config), vp, C.POINTER(ltg_batch), C.POINTER(ltg_batch), vp, vp, vp]),
    "ltg_rank_finish": (C.c_int, [C.POINTER(ltg_batch), vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]),
    "ltg_fp8_roundtrip": (C.c_int, [vp, vp, C.c_int32, vp]),
    "ltg_debug_split": (C.c_int, [vp, vp, C.c_int32, vp]),
    "ltg_debug_gemm": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]),
}

_lib = None


class LtgError(RuntimeError):
    pass


def load():
    """dlopen libltg_hip.so and bind every declared symbol.  Raises (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LtgError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C long-tail-gan_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # LTG_HIP_LIB alone points at another build of THIS library: same strict checks.  LTG_AB_COMPAT=1 (scripts/ab_libs.sh, A/B timing
    # against an OLDER build): entry points it lacks stay unbound (Engine.sharded_step_ok is then False) and ABI 9 / 10 are accepted --
    # struct layouts only ever grew.
    ab = bool(os.environ.get("LTG_HIP_LIB")) and os.environ.get("LTG_AB_COMPAT", "0") == "1"
    for name, (res, args) in SYMBOLS.items():
        if ab and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.ltg_abi_version()
    if got != LTG_ABI_VERSION and not (ab and 9 <= got < LTG_ABI_VERSION):
        raise LtgError("ABI version mismatch: library %d, binding %d" % (got, LTG_ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise LtgError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))
