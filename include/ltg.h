/*
 * ltg.h -- C ABI of the MI355X-native Long-Tail-GAN training path (libltg_hip.so).
 *
 * The reference (ash-shar/Long-Tail-GAN) has no FFI of its own: its execution boundary is the
 * five TensorFlow `sess.run` call sites of Codes/train.py / Codes/test.py.  Each entry point below
 * replaces one of them (cited per function).  Conventions:
 *
 *   - plain C: pointers + sizes, no torch / C++ types; every pointer is a DEVICE pointer unless
 *     the parameter name starts with `host_`;
 *   - the caller owns every buffer (parameters, Adam moments, activations, workspace); the library
 *     allocates nothing, keeps no global state and is asynchronous on the given HIP stream;
 *   - return value: 0 on success, a negative LTG_E* code on error; never throws;
 *   - randomness: every random tensor can be injected through an optional pointer; NULL selects the
 *     on-device counter RNG keyed by (cfg->seed, stream id, rng_step, element index), which the CPU
 *     oracle reproduces bit-for-bit (oracle/ltg_oracle.py rng_*).
 *
 * Layouts in HBM (all row-major, fp32 unless stated):
 *   W_q0  [I][H]      encoder layer 0   (TF shape [I,600],  MultiVAE.py:199)
 *   W_q1  [H][2Z]     encoder layer 1   (TF shape [600,400])
 *   W_p0  [Z][H]      decoder layer 0   (TF shape [200,600])
 *   W_p1t [I][H]      decoder layer 1 stored ITEM-MAJOR (transpose of TF's [600,I], MultiVAE.py:218)
 *   b_q0 [H], b_q1 [2Z], b_p0 [H], b_p1 [I]
 *   discriminator: emb [F][h0] (frozen), w1 [h0][h1], w2 [h0][h2], w3 [h1+h2][h3], w4 [h3], b*.
 */
#ifndef LTG_H
#define LTG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTG_ABI_VERSION 14

#define LTG_OK 0
#define LTG_EINVAL (-1)     /* bad argument (NULL pointer, negative size, unsupported dims) */
#define LTG_EWORKSPACE (-2) /* workspace too small */
#define LTG_ELAUNCH (-3)    /* HIP launch error (hipGetLastError != hipSuccess) */

#define LTG_PREC_BF16 0 /* decoder GEMM operands rounded to bf16, fp32 accumulate (MFMA 16x16x32 bf16) */
#define LTG_PREC_FP32 1 /* exact fp32 MFMA (16x16x4 f32) everywhere */
#define LTG_PREC_FP8 2  /* d_precision only: OCP e4m3 operands, static power-of-two scales, fp32 accumulate (16x16x32 fp8) */

#define LTG_DARITH_FP32 0   /* ltg_config.d_arith: exact fp32 MFMA (16x16x4 f32) */
#define LTG_DARITH_BF16X6 1 /* three-way bf16 split of every fp32 operand, six cross terms on the bf16 matrix pipe: fp32-accurate */
#define LTG_DARITH_BF16X4 2 /* two-way split, four cross terms: 2^-17 per product, opt-in */

typedef void* ltg_stream; /* hipStream_t */

/* Model / optimiser configuration.  Codes/config.ini:2-12, Codes/generator.py:13-18,
 * Codes/train.py:160 (AdamOptimizer defaults beta1=.9 beta2=.999 eps=1e-8). */
typedef struct ltg_config {
    int32_t n_items; /* I */
    int32_t h_enc;   /* H = p_dims[1] = 600 */
    int32_t z_dim;   /* Z = p_dims[0] = 200 */
    int32_t d_feat;  /* FEATURE_LEN: rows of the frozen embedding table */
    int32_t d_h0, d_h1, d_h2, d_h3;
    int32_t precision; /* LTG_PREC_* */
    /* kernel selection knob, 0 = auto; every setting computes the same function (ABI v14: was `reserved0`).
     * STABLE bits -- the host layer and the parity tests select code paths with them:
     *   bit 18  the generic (round-1, LDS-staged, any size) kernels instead of the latency path of csrc/ltg_fast.h
     *   bit 17 / 26  the streaming decoder forward: its second form (h2 resident in LDS, per-wave item tiles; default from 65 536 items) from
     *           8 192 items / its first form at every size
     *   bit 19  fp8 discriminator: the backward converts its operands on the fly (instead of operand-format storage)
     *   bit 20  the forward-only tower as three launches (fks_d_l1 -> fks_d_l2 -> fk_d_y) instead of the one-kernel tower of csrc/ltg_tower.h
     *   bit 21 / 22  softmax statistics from a second pass over the logits / fp32 instead of bf16 dlogits on the streaming path
     * MEASUREMENT bits -- A/B switches of closed or running experiments (profiles/README.md); they may go without an ABI bump:
     *   bits 0-3 decoder weight-gradient kernel (1..4 = tile variant of the generic kernel, 9 = no streaming kernels); bit 6 no fork of the D
     *   step's backward jobs; bit 9 no fake-tower fork; bits 10-12 discriminator tiles (1 scalar loaders, 2 all 64 x 64, 3 all 128 x 128);
     *   bit 13 / 15 / 16 scalar instead of 16-byte loaders (middle layers / embedding gathers / small-item decoder kernels); bit 14 column-blocked
     *   sparse W_q0 gradient; bit 24 fp8 register-resident forward tiles; bit 25 sparse gradient and Adam as two launches; bits 27-30 = k: the
     *   streaming weight update with 256 - 8 k workgroups.
     * (bit 23, the register-resident forward-only towers, is gone: the one-kernel tower superseded the experiment.) */
    int32_t tuning;
    /* item shard of this rank: it owns global items [item_lo, item_lo + n_items); n_items_global = 0 means
     * unsharded (n_items_global = n_items, item_lo = 0).  W_q0 / W_p1t / b_p1 and their Adam moments hold
     * only the local rows; CSR indices are local; fake-pair and candidate ids stay global. */
    int32_t item_lo;
    int32_t n_items_global;
    /* operand precision of the discriminator GEMMs (forward and backward; discriminator.py:16-55 is fp32):
     * LTG_PREC_FP32 = the reference's arithmetic (default of the host layer), LTG_PREC_BF16, LTG_PREC_FP8
     * (BASELINE config 5).  Accumulation, activations, loss, Adam and the master weights are fp32 in every mode. */
    int32_t d_precision;
    /* arithmetic of the fp32 discriminator's GEMMs (d_precision = LTG_PREC_FP32, config.ini-sized layers; discriminator.py:23-55, train.py:142-143,163):
     * LTG_DARITH_FP32 = v_mfma_f32_16x16x4_f32, the exact fp32 fma chain; LTG_DARITH_BF16X6 = every fp32 operand split into three bf16 terms
     * (x = hi + mid + lo exactly) and the six leading cross terms multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- leaves out
     * 2^-26 |a b| per product, below fp32's own rounding of it; LTG_DARITH_BF16X4 = two terms per operand, four cross terms (2^-17 per product:
     * NOT fp32-accurate, opt-in).  Bits 4-7 (measurements only): which of the step's four GEMM kernels take the split form (l1, l2, bwd1, bwd2;
     * 0 = the library's choice). */
    int32_t d_arith;
    float lr, beta1, beta2, adam_eps;
    uint64_t seed;
} ltg_config;

/* Generator parameters + Adam moments, reference order (MultiVAE.py:129-141):
 * 0 W_q0, 1 W_q1, 2 W_p0, 3 W_p1t, 4 b_q0, 5 b_q1, 6 b_p0, 7 b_p1. */
typedef struct ltg_gen_state {
    float* p[8];
    float* m[8];
    float* v[8];
    /* optional bf16 shadow of W_p1t: [n_items][608] bf16 (row = item, K zero-padded to 608), rebuilt by
     * ltg_refresh_shadow and kept in step by the Adam epilogue of the G step.  When present (and n_items >=
     * 8192, n_items % 8 == 0, bf16 precision, <= 128 rows) the decoder forward and the dh2 product stream it
     * (1216 B/item instead of 2400 B/item).  NULL: the fp32 rows are converted on the fly. */
    uint16_t* wp1t_bf16;
    /* optional lazy Adam clock of W_q0 (item slabs of 8192 items or more).  tf.train.AdamOptimizer (train.py:160-164) moves
     * every row of W_q0 every step, but a row's gradient is zero unless one of the batch's users holds the item, and a
     * zero-gradient step is a function of (W, m, v, lr_t) of that row alone: it can be applied LATER, step by step, with the
     * same arithmetic -- before the row is next read.  q0_last[i] = ordinal of the last G step applied to row i; q0_ord =
     * ordinal of the last G step issued (HOST value: the caller adds 1 after every ltg_g_step / ltg_g_bwd_rest);
     * q0_lr_hist[j % LTG_Q0_HIST] = lr_t of G step j (written by the library).  Every G step applies itself to the batch's
     * rows (caught up first), and catches up the rows i = ord (mod q0_period), so that no row lags more than q0_period steps
     * (1 <= q0_period <= LTG_Q0_HIST / 2).  Every forward catches up the rows it reads.  ltg_g_flush brings all rows to
     * q0_ord (before W_q0 / its moments are read by anything else: checkpoints, host copies).  Results are bit-identical
     * to the dense sweep.  q0_last == NULL: dense sweep every step. */
    int32_t* q0_last;
    float* q0_lr_hist;
    int32_t q0_ord;
    int32_t q0_period;
} ltg_gen_state;
#define LTG_Q0_HIST 1024

/* Discriminator: emb is read-only (discriminator.py:14 is not in d_params, :47).
 * Trainable order (discriminator.py:47): 0 w1, 1 b1, 2 w2, 3 b2, 4 w3, 5 b3, 6 w4, 7 b4. */
typedef struct ltg_disc_state {
    const float* emb;
    float* p[8];
    float* m[8];
    float* v[8];
    /* optional OPERAND-FORMAT shadows for d_precision = LTG_PREC_FP8 (all four or none; every layer size a multiple of 64):
     * OCP e4m3 bytes with the mode's static scales (embeddings and weights 2^8), k-contiguous for the forward GEMMs --
     * emb_fp8 [F][h0], w1t_fp8 [h1][h0], w2t_fp8 [h2][h0], w3t_fp8 [h3][h1+h2] (the weight shadows TRANSPOSED).  Built by
     * ltg_refresh_d_shadow, kept in step by the Adam sweep of ltg_d_step / ltg_d_apply.  When present the two forward layers
     * read 1 byte per operand element instead of converting 4-byte values on the fly. */
    const uint8_t* emb_fp8;
    uint8_t* w1t_fp8;
    uint8_t* w2t_fp8;
    uint8_t* w3t_fp8;
    /* optional fifth shadow (ABI v10): w3 in its OWN layout [h1+h2][h3] in e4m3 -- the operand of the backward product
     * dpre1 = dpre3 . w3^T, which contracts over h3.  With it (and h0, h1+h2, h3 multiples of 128, h1, h2 multiples of 64) every
     * backward GEMM of the step reads e4m3 bytes in operand format as well (csrc/ltg_fp8bwd.h); without it they convert fp32
     * operands on the fly.  Same values either way. */
    uint8_t* w3_fp8;
} ltg_disc_state;

/* A batch of user rows in CSR form (replaces the dense [B,I] float32 feed of train.py:194-198).
 * indptr holds absolute offsets into indices/values.  The transposed view (the batch's entries grouped
 * by item) is only required by the G step (sparse gradient of W_q0):
 *   uitem[u]  = u-th distinct item of the batch (ascending),  uptr[u..u+1] = its entry range,
 *   rowidx/csr_pos = per entry: local row, index into indices[];  slot[i] = u or -1 for every item i. */
typedef struct ltg_batch {
    int32_t n_rows;
    int32_t n_unique;       /* number of distinct items in the batch */
    const int32_t* indptr;  /* [n_rows+1] */
    const int32_t* indices; /* item ids, ascending within a row */
    const float* values;    /* NULL => 1.0f */
    const int32_t* slot;    /* [n_items] item -> index into uitem / the gradient rows, or -1.  May be NULL: the library then builds the
                             * map of the batch in its workspace each step (no n_batches x n_items cache on the caller's side) */
    const int32_t* uptr;    /* [n_unique+1] offsets into rowidx/csr_pos (relative to 0) */
    const int32_t* rowidx;  /* local row of each transposed entry */
    const int32_t* csr_pos; /* index of that entry in indices[] */
    const float* row_norm2; /* optional [n_rows]: sum x^2 over the FULL row (needed when the items are sharded) */
    const int32_t* uitem;   /* optional [n_unique] (ABI v11): the distinct items themselves, ascending.  The same value as
                             * indices[csr_pos[uptr[u]]] -- with it the kernels that walk the distinct items (catch-up of the lazy Adam
                             * clock, sparse W_q0 gradient) learn the item in ONE dependent load instead of three and can request its
                             * weight / moment rows before the gather has returned */
} ltg_batch;

/* Activations of one generator forward; caller-owned, sizes for n_rows rows.
 * logits/lse are what the sampler, the G step and the metrics consume. */
typedef struct ltg_gen_acts {
    float* h1;      /* [n_rows][H]   tanh(enc0) */
    float* mulv;    /* [n_rows][2Z]  mu | logvar */
    float* z;       /* [n_rows][Z] */
    float* h2;      /* [n_rows][H]   tanh(dec0) */
    float* logits;  /* [n_rows][I] */
    float* lse;     /* [n_rows]      log-sum-exp of the row */
    float* kl_rows; /* [n_rows]      per-row KL (MultiVAE.py:161) */
    float* row_scale; /* [n_rows]    1/(keep*||x||_2) */
} ltg_gen_acts;

/* Optional timing probe: the library records the two HIP events (hipEvent_t) around the launch of the
 * kernel `kernel_id` on the call's stream.  Measurement only; NULL = off. */
#define LTG_K_ENC0_FWD 1
#define LTG_K_ENC1 2
#define LTG_K_DEC0 3
#define LTG_K_DEC1_FWD 4
#define LTG_K_D_L1 5
#define LTG_K_D_L2 6
#define LTG_K_D_BWD1 7
#define LTG_K_D_BWD2 8
#define LTG_K_D_ADAM 9
#define LTG_K_DH2 10
#define LTG_K_DEC1_BWD_ADAM 11
#define LTG_K_ENC0_BWD_ADAM 12
#define LTG_K_DZ 13
#define LTG_K_DH1 14
#define LTG_K_WGRAD_P0 15
#define LTG_K_WGRAD_Q1 16
#define LTG_K_ROW_DLOGITS 17 /* softmax statistics + losses + dlogits of a row (small item slabs) */
#define LTG_K_ENC0_GRAD 18   /* sparse gradient rows of W_q0 */
#define LTG_K_G_TAIL 19      /* the generator's Adam updates as jobs of one launch */
#define LTG_K_EXCH_H1 20      /* ltg_g_step_sharded: the three in-stream exchanges (events around the collective call on the step's stream) */
#define LTG_K_EXCH_ROWPART 21
#define LTG_K_EXCH_DH2 22
#define LTG_K_COUNT 23
typedef struct ltg_probe {
    int32_t kernel_id;
    int32_t reserved0;
    void* ev_start;
    void* ev_stop;
} ltg_probe;

typedef struct ltg_fwd_opts {
    float keep_prob;   /* keep_prob_ph, default 0.75 (MultiVAE.py:31) -- ON at inference too (Q3) */
    float is_training; /* is_training_ph (MultiVAE.py:101) */
    uint64_t rng_step; /* counter for the on-device RNG */
    const uint8_t* drop_keep; /* optional keep flags, indexed like indices[] (drop_keep[e] belongs to indices[e]) */
    const float* eps;         /* optional [n_rows][Z] */
    const ltg_probe* probe;   /* optional */
    /* ltg_vae_forward over SEVERAL batches at once (phase C / evaluation need no weight update in between): rows
     * [k * rows_per_step, (k + 1) * rows_per_step) draw their dropout with counter rng_step + k and row index r - k * rows_per_step,
     * i.e. exactly what k separate calls with consecutive rng_step values draw.  0 = one batch.  Needs is_training == 0. */
    int32_t rows_per_step;
    int32_t reserved0;
} ltg_fwd_opts;

/* Fake/real (popular, niche) id pairs.  Rows with id < 0 are holes (dropped pairs, Q9/Q10). */
typedef struct ltg_pairs {
    int32_t n;              /* number of slots */
    int32_t reserved0;
    const int32_t* pop;     /* [n] popular item id */
    const int32_t* niche;   /* [n] niche / generated item id */
    const int32_t* row;     /* [n] local user row of the pair (fake pairs; may be NULL for real) */
} ltg_pairs;

typedef struct ltg_d_opts {
    float keep_prob;    /* 0.7 (train.py:300) */
    int32_t adam_t;     /* shared Adam step AFTER this update (t >= 1), Q5 */
    uint64_t rng_step;
    const uint8_t* drop_real[3]; /* optional keep flags [n_real][h1], [n_real][h2], [n_real][h3] */
    const uint8_t* drop_fake[3];
    const ltg_probe* probe; /* optional */
    /* optional (ABI v12; ltg_d_step at the fp32 default sizes): jobs B / C of the backward's first stage -- dw3, db3, dw4, db4, d_loss: they
     * need the forward only -- run on `aux_stream` beside the critical chain job A -> stage 2, handed over through device words like
     * ltg_pipe's: sync = 4 zeroed words owned by the caller (0: forward complete, 1: jobs B / C ended, 2: polls that gave up = poison:
     * the Adam sweep then returns at once and the caller must treat it as fatal), seq = the call's ordinal on these words (+ 1 per
     * call, starting at 1).  The two streams must run concurrently (ltg_g_pipe_probe tests a pair).  NULL: one stream, five launches. */
    ltg_stream aux_stream;
    uint32_t* sync;
    uint32_t seq;
    int32_t reserved0;
} ltg_d_opts;

typedef struct ltg_g_opts {
    ltg_fwd_opts fwd;   /* keep 0.75, is_training 1 (train.py:326) */
    float anneal;       /* anneal_ph */
    float gan_lambda;   /* gen_lambda */
    float d_keep_prob;  /* 0.7 */
    int32_t adam_t;     /* shared Adam step AFTER this update */
    uint64_t d_rng_step;
    const uint8_t* drop_fake[3]; /* optional */
    const int32_t* cnt;  /* device scalar: sampled_cnt (number of valid fake pairs) */
    const ltg_probe* probe; /* optional (the forward part uses fwd.probe) */
    /* optional fork/join: run the fake-tower forward on aux_stream concurrently with the generator forward.
     * ev_fork / ev_join are caller-created hipEvent_t (the library allocates nothing); all three or none. */
    ltg_stream aux_stream;
    void* ev_fork;
    void* ev_join;
    /* item-sharded runs: ltg_g_bwd_dec1 already updated W_p1t / b_p1 of this step (issued while the dh2 all-reduce was in
     * flight), ltg_g_bwd_rest must not do it again */
    int32_t dec1_done;
    /* item-sharded runs: ltg_g_fake_tower already left the fake tower's forward in the workspace (issued on another stream
     * beside the generator forward and its exchanges), ltg_g_bwd_dec must not run it again.  0 or 1. */
    int32_t fake_done;
    /* optional third caller-created hipEvent_t (with aux_stream): ltg_g_step then runs the rotating slice of the lazy Adam
     * clock of W_q0 (arithmetic-bound) on aux_stream, beside the HBM-bound decoder kernels, and joins it at the end */
    void* ev_sweep;
    /* optional y_generated of this batch's fake pairs [fake->n], computed ahead by ltg_fake_tower_batched with this step's
     * d_rng_step: the step then runs no fake tower at all */
    const float* y_pre;
} ltg_g_opts;

/* Static per-user sampling inputs for a batch (results of the index path, data_processing.py). */
typedef struct ltg_sample_inputs {
    int32_t n_rows;
    int32_t max_cand;         /* max candidate-set length in this batch (sizes the LDS key buffer; <= 16384) */
    const int32_t* cand_ptr;  /* [n_rows+1]  USER_TAGS_TO_SAMPLE (data_processing.py:170-224) */
    const int32_t* cand_idx;  /* ascending candidate item ids */
    const int32_t* pop_ptr;   /* [n_rows+1]  user's popular list in file order (data_processing.py:72-96) */
    const int32_t* pop_idx;
    const int32_t* n_sample;  /* [n_rows] to_sample = len(user niche list); 0 => invalid user (Q8) */
    const int32_t* slot_ptr;  /* [n_rows+1] prefix sum of n_sample: output slot range per user */
    const uint8_t* valid_item;/* [I] 1 if the id is in ITEM_FEATURE_DICT (Q9) */
    uint64_t rng_step;
    const float* u_gumbel;    /* optional [n_cand total] uniforms aligned with cand_idx */
    const float* u_pick;      /* optional [n_slots] uniforms aligned with slots */
    const float* cand_logit;  /* optional, aligned with cand_idx: replaces the [B,I] logits (item-sharded runs) */
    /* several batches in one call (see ltg_fwd_opts.rows_per_step): rows of group k = r / rows_per_step draw with counter
     * rng_step + k and row index r % rows_per_step, and add their valid pairs to cnt_out[k].  0 = one batch. */
    int32_t rows_per_step;
    int32_t reserved0;
} ltg_sample_inputs;

int32_t ltg_abi_version(void);

/* Bytes of workspace needed by any entry point for at most max_rows user rows and max_pairs
 * discriminator rows (real+fake) per call. */
size_t ltg_workspace_bytes(const ltg_config* cfg, int32_t max_rows, int32_t max_pairs);
/* Bytes of the optional scratch of ltg_vae_forward (its ws argument) for at most max_rows rows: row statistics of large item
 * slabs in one pass.  Without it (ws == NULL or smaller) the forward computes them row by row. */
size_t ltg_forward_scratch_bytes(const ltg_config* cfg, int32_t max_rows);

/* Generator forward: replaces sess.run(generator_out, {input_ph: X}) -- Codes/train.py:200, :339,
 * Codes/test.py:146 (graph: MultiVAE.py:145-186, softmax :143).  Fills `acts`; if probs_out != NULL
 * also writes softmax probabilities [n_rows][I] (Q1). */
int ltg_vae_forward(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch,
                    const ltg_fwd_opts* opts, const ltg_gen_acts* acts, float* probs_out,
                    void* ws, size_t ws_bytes, ltg_stream stream);

/* Fake-pair sampling for one batch: replaces the Python loop Codes/train.py:212-251 including
 * sample_from_generator_new (Codes/sample.py:40-67).  Reads logits/lse of ltg_vae_forward.
 * Writes gen_out/pop_out [n_slots] (-1 = hole) and cnt_out[0] = number of valid pairs. */
int ltg_sample_pairs(const ltg_config* cfg, const ltg_sample_inputs* in, const float* logits,
                     const float* lse, int32_t* gen_out, int32_t* pop_out, int32_t* cnt_out,
                     ltg_stream stream);

/* Discriminator update: replaces sess.run([d_trainer, d_loss_mean]) -- Codes/train.py:300
 * (graph discriminator.py:3-58, loss train.py:142, Adam train.py:160-163).
 * loss_out[0] = d_loss (device). */
int ltg_d_step(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* real,
               const ltg_pairs* fake, const ltg_d_opts* opts, float* loss_out, void* ws,
               size_t ws_bytes, ltg_stream stream);

/* ---- The discriminator step cut at its exchange point, for multi-GPU runs that split the PAIR ROWS over ranks:
 * the logical pair batch is the concatenation real | fake (rows 0 .. real->n + fake->n); a rank runs forward + backward over
 * rows [row_lo, row_hi) with the full step's dropout draw (the counter RNG is indexed by the global row) and gets ONE
 * gradient vector: ltg_d_grad_floats(cfg) floats = the eight trainable tensors in discriminator.py:47 order back to back
 * (w1, b1, w2, b2, w3, b3, w4, b4), then this rank's share of d_loss (train.py:142), padded to whole float4.  The host
 * all-reduces (sum) that vector with RCCL; ltg_d_apply then runs the identical TF-Adam sweep on every rank and writes
 * d_loss.  ltg_d_step == ltg_d_grad over all rows + ltg_d_apply.  opts->adam_t is ignored by ltg_d_grad. */
size_t ltg_d_grad_floats(const ltg_config* cfg);
int ltg_d_grad(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* real, const ltg_pairs* fake, int32_t row_lo,
               int32_t row_hi, const ltg_d_opts* opts, float* grad_out, void* ws, size_t ws_bytes, ltg_stream stream);
int ltg_d_apply(const ltg_config* cfg, const ltg_disc_state* disc, const float* grad, int32_t adam_t, float* loss_out,
                ltg_stream stream);

/* Generator update: replaces sess.run([g_trainer, g_loss_mean, g_vae_loss, gan_loss]) --
 * Codes/train.py:326 (losses train.py:145-157, Adam :164).  The dense mask feed `generated_tags`
 * is replaced by the (row, gen id) list of `fake`.  loss_out (device, >= 8 floats): [0..2] = g_loss,
 * vae_loss, gan_loss; [3] = sum_S p, [4] = sum_j y_j, [5] = c = lambda/cnt * sum_j y_j. */
int ltg_g_step(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc,
               const ltg_batch* batch, const ltg_pairs* fake, const ltg_g_opts* opts,
               const ltg_gen_acts* acts, float* loss_out, void* ws, size_t ws_bytes,
               ltg_stream stream);

/* ---- The generator step cut at its three exchange points, for item-sharded multi-GPU runs (one process per
 * GPU; the collectives between the stages are issued by the host with RCCL, the library itself never
 * communicates).  ltg_g_step == the four stages back to back with n_ranks = 1.
 *   ltg_g_fwd_enc   enc-0 over the local item slab -> acts->h1 = partial PRE-activation   [all-reduce sum h1]
 *   ltg_g_fwd_rest  bias+tanh, enc-1, reparam, dec-0, local logits, row partials [n_rows][5] [all-gather]
 *   ltg_g_bwd_dec   combine partials (lse, losses), fake tower, dlogits, local dh2 [n_rows][H] [all-reduce sum]
 *   ltg_g_bwd_rest  Adam on the local W_p1t/b_p1 rows, replicated middle layers, local W_q0 rows
 * ltg_rowstats_combine: lse of the full rows from all-gathered partials (phase C / evaluation).
 * ltg_gather_cand_logits: this rank's candidate logits, 0 elsewhere (all-reduce -> ltg_sample_inputs.cand_logit). */
int ltg_g_fwd_enc(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_fwd_opts* opts,
                  const ltg_gen_acts* acts, ltg_stream stream);
int ltg_g_fwd_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_pairs* fake,
                   const ltg_fwd_opts* opts, const ltg_gen_acts* acts, float* rowpart_out, void* ws, size_t ws_bytes,
                   ltg_stream stream);
int ltg_rowstats_combine(const ltg_config* cfg, const float* rowpart_all, int32_t n_ranks, int32_t n_rows, float* lse_out,
                         void* ws, size_t ws_bytes, ltg_stream stream);
int ltg_g_bwd_dec(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* batch,
                  const ltg_pairs* fake, const ltg_g_opts* opts, const ltg_gen_acts* acts, const float* rowpart_all,
                  int32_t n_ranks, float* loss_out, float* dh2_out, void* ws, size_t ws_bytes, ltg_stream stream);
int ltg_g_bwd_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_pairs* fake,
                   const ltg_g_opts* opts, const ltg_gen_acts* acts, const float* dh2, void* ws, size_t ws_bytes,
                   ltg_stream stream);
/* The part of ltg_g_bwd_rest that does not need the all-reduced dh2: Adam on the local W_p1t / b_p1 rows (reads dlogits and
 * h2 of ltg_g_bwd_dec).  Issue it right after starting the dh2 all-reduce, then call ltg_g_bwd_rest with opts->dec1_done = 1. */
/* The fake tower's forward of a G step (replicated on every rank; needs only the fake pairs and the discriminator) into the
 * workspace ltg_g_bwd_dec reads it from: same cfg / n_rows / fake / ws as that call, any stream -- the caller orders it
 * after the previous step's ltg_g_bwd_dec and before this step's, and sets ltg_g_opts.fake_done = 1. */
int ltg_g_fake_tower(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* fake, const ltg_g_opts* o, int32_t n_rows, void* ws,
                     size_t ws_bytes, ltg_stream stream);
/* The fake tower's forward (discriminator.py:52-55 on x_popular_g / x_generated; the only part of the discriminator the g_trainer
 * fetch needs, train.py:326) for MANY pair batches in one pass -- the discriminator does not move during phase G (train.py:307-329)
 * and the fake pairs are fixed for the epoch, so every y_generated of a sub-epoch is known before its first G step.
 * fake = the concatenated slots of the batches; seg_of[fake->n] = index s of the batch each slot belongs to, seg_row0[s] = first
 * slot of batch s (the dropout RNG sees a slot's row INSIDE its batch), seg_step[s] = the d_rng_step the G step of batch s will
 * use: the same draws, the same bits as the tower inside ltg_g_step.  y_out[fake->n] (0 for holes); the G steps take their slice
 * through ltg_g_opts.y_pre.  ws: ltg_workspace_bytes(cfg, 1, fake->n). */
int ltg_fake_tower_batched(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* fake, const int32_t* seg_of,
                           const int32_t* seg_row0, const uint64_t* seg_step, float d_keep_prob, float* y_out, void* ws, size_t ws_bytes,
                           ltg_stream stream);
int ltg_g_bwd_dec1(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_pairs* fake,
                   const ltg_g_opts* opts, const ltg_gen_acts* acts, void* ws, size_t ws_bytes, ltg_stream stream);
int ltg_gather_cand_logits(const ltg_config* cfg, const ltg_sample_inputs* in, const float* logits, float* out,
                           ltg_stream stream);

/* ---- The item-sharded generator step as ONE call (ABI v10): every launch of the step AND its three exchanges issued in-stream,
 * with the two off-critical-path pieces of step t running beside step t + 1.  Replaces the five-call sequence ltg_g_fwd_enc ..
 * ltg_g_bwd_rest (one sess.run per step in the reference, train.py:326) for the configuration every large item slab runs in:
 * bf16 decoder operands with the W_p1t shadow, the lazy Adam clock of W_q0, n_items >= 8192, n_items % 8 == 0, <= 128 rows
 * (ltg_g_step_sharded_ok tells; other configurations use the cut-point calls above).
 *
 * Collectives: the library links no communication library.  The caller hands in its communicator and the two entry points with
 * the signatures of ncclAllReduce / ncclAllGather (rccl.h:611, :678; enums passed as int: LTG_NCCL_FLOAT32 = ncclFloat32,
 * LTG_NCCL_SUM = ncclSum) -- on a GPU node the addresses of RCCL's own functions, so the exchanges are RCCL kernels on the
 * caller's stream between the library's kernels (no host round trip); a test rig may pass host functions that synchronise
 * the stream and reduce through another backend.  comm == NULL: no exchange (one GPU); a communicator of ONE rank still gets its
 * three calls (the single-GPU proxy of a rank's step measures them).
 *
 * Pipeline (ltg_pipe: ONE caller-created side stream, two events and the exchange buffers; the library allocates nothing).
 * One fork per step, behind the dh2 product of step t (the last reader of the W_p1t shadow), puts onto the side stream the Adam update
 * of the local W_p1t / b_p1 rows of step t (HBM-bound, the largest kernel; needs only dlogits and h2): it runs beside the rest of
 * step t's backward and the encoder half of step t + 1's forward.  The rotating slice of the lazy Adam clock that step t - 1 owes (rows
 * i = ord (mod q0_period) up to ordinal ord = t - 1) runs on the side stream too, between the catch-up of call t and that of call t + 1.
 *
 * Hand-over between the two streams: a cross-stream event pair costs ~12 us per direction on this hardware and ~6 us of bubble on the
 * recording stream (scripts/micro/sync_cost.hip), so the streams meet through words of device memory (ltg_pipe.sync; `seq` = the
 * call's ordinal) -- no packet waits on the caller's stream at all:
 *   word 0  the dh2 product of call seq is complete (stored by the slab sum when it starts; a one-wave kernel in front of the weight
 *           update polls for it)
 *   word 1  the weight update of call seq has READ h2 -- it reads h2 in its prologue only: its workgroups count themselves in word 8
 *           behind it and the last one stores the ordinal (ragged slabs: stored when the update ends).  dec-0 of call seq + 1 overwrites h2.
 *   word 7  the weight update of call seq has ENDED (one wave behind it): the streaming forward of call seq + 1 reads the shadow rows
 *           and the bias it writes
 *   word 5  enc-0 of call seq has started = the catch-up of the batch's rows is complete (a one-wave kernel in front of the slice polls)
 *   word 6  the clock's work beside call seq -- its slice and, ABI v13, the catch-up of the next batch's rows -- has ended (stored by the
 *           next kernel of the side stream when it starts): the catch-up of call seq + 1 must not meet it on a row
 *   word 9  dh1 of call seq is complete (stored by the sparse gradient kernel when it starts; a one-wave kernel in front of the Adam
 *           tail on ltg_pipe.tail_stream polls);  word 10  that tail has ended (one wave behind it): enc-1 of call seq + 1 reads what it updates
 *   word 2  polls that gave up (each is bounded: 30 s) = the pipe's POISON
 * Every poll is made by ONE thread: a one-wave kernel of the side stream, or -- on the caller's stream -- the last thread of the kernel
 * IN FRONT of the one that needs the word (enc-1 for word 1, dec-0 for word 7, enc-0 for word 10, the step's last kernel for word 6): stream order then
 * gates the consumer whatever its size, and no workgroup of a large launch holds a CU while it waits for a producer that may still
 * need one.  Once word 2 is non-zero every kernel of the step that writes h2 or the model returns at once, so the model stays what it
 * was when the wait expired; the caller checks word 2 before it reads the model out (end of a phase, checkpoint) and treats it as fatal.
 * LTG_PIPE_SLICE_IN_TOUCH keeps the slice in the catch-up launch of call t (one launch over the batch's rows and the slice's rows; a
 * row in both sets goes to whichever workgroup's atomic max on its clock comes first).  LTG_PIPE_EVENTS (or sync == NULL) selects event
 * pairs instead of words: stream waits on events recorded by the PREVIOUS call (a never-recorded event does not block), the slice in
 * the catch-up launch.  A pipe is used in ONE mode between two joins, seq increases by 1 per call; the two streams of the device-word
 * mode must be concurrent (ltg_g_pipe_probe).  After a call that FAILED the caller synchronises, zeroes the words and restarts seq at 0.
 * Before anything else reads W_p1t / W_q0 / their moments (ltg_g_flush, ltg_vae_forward, a checkpoint) the caller runs
 * ltg_g_pipe_join on the stream that will read them.  Results equal ltg_g_step's / the cut-point sequence's bit for bit (same
 * kernels, same order of additions; the slice only runs later).
 *
 * Catch-up AHEAD (ABI v13; device-word mode with the slice on the side stream, batches that carry `uitem`, h_enc <= 768): the catch-up
 * of the lazy clock -- the rows of W_q0 the batch reads, brought up to the caller's clock before enc-0 -- is the FIRST kernel of the
 * step's critical chain and moves 24 B per parameter of every distinct row.  A caller that knows the NEXT batch sets
 * ltg_pipe.next_uitem / next_nu: call t then brings those rows up to ordinal t ON THE SIDE STREAM, behind its slice (so word 6 covers
 * it), beside its own forward -- every row except the ones batch t itself holds (those get step t's gradient from the sparse gradient
 * kernel and are at t with it).  Which rows batch t holds is read from ltg_pipe.q0_mark (one int32 per local item, zeroed once by the
 * caller and with every reset of seq): the catch-up -- or the ahead kernel of the call before -- stores the call's ordinal there.  The
 * zero-gradient steps are the same expressions in the same order, only earlier: results do not change by a bit.  Call t + 1 on the
 * SAME batch the pipe announced, with nothing else having moved gen->q0_ord in between, sets ltg_pipe.caught_up = 1 and the library
 * launches no catch-up.  ltg_g_step_sharded_plan says whether a call with these arguments runs the ahead kernel. */
#define LTG_NCCL_FLOAT32 7
#define LTG_NCCL_SUM 0
typedef struct ltg_comm {
    void* comm;      /* ncclComm_t */
    int32_t n_ranks; /* >= 1 */
    int32_t rank;
    int (*all_reduce)(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, ltg_stream stream);
    int (*all_gather)(const void* sendbuf, void* recvbuf, size_t sendcount, int dtype, void* comm, ltg_stream stream);
} ltg_comm;

/* One-shot exchange over peer-mapped staging buffers: a second transport with ltg_comm's entry-point signatures (csrc/ltg_oneshot.h;
 * correctness only -- RCCL is the transport of every measurement).  The caller allocates ltg_oneshot_stage_bytes() bytes of ZEROED device
 * memory per rank, shares them between the ranks (hipIpcGetMemHandle / hipIpcOpenMemHandle: stage[q] = rank q's buffer as mapped into this
 * process, stage[rank] = its own), and puts &ltg_oneshot into ltg_comm.comm with ltg_oneshot_all_reduce / ltg_oneshot_all_gather as the
 * entry points.  HOST structure: the library advances `seq` with every exchange (all ranks issue the same sequence of exchanges); counts up
 * to max_floats (all-gather: per rank); float32 / sum only.  Replaces nothing in the reference (Codes/train.py has no distributed code). */
#define LTG_ONESHOT_MAX_RANKS 16
typedef struct ltg_oneshot {
    int32_t n_ranks;
    int32_t rank;
    uint32_t seq;      /* exchanges issued so far; 0 at start */
    uint32_t limit_ms; /* bound of a device-side wait for a peer's message (0 = 30 s); a wait that gives up counts in `expired` of the stage */
    size_t max_floats;
    void* stage[LTG_ONESHOT_MAX_RANKS];
} ltg_oneshot;
size_t ltg_oneshot_stage_bytes(int32_t n_ranks, size_t max_floats);
/* byte offset of the `expired` counter (uint32) inside a rank's stage: the host reads it after a synchronisation */
size_t ltg_oneshot_expired_offset(int32_t n_ranks);
int ltg_oneshot_all_reduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, ltg_stream stream);
int ltg_oneshot_all_gather(const void* sendbuf, void* recvbuf, size_t sendcount, int dtype, void* comm, ltg_stream stream);

typedef struct ltg_pipe {
    ltg_stream side_stream;
    void* ev_fork; /* hipEvent_t x 2, timing disabled (LTG_PIPE_EVENTS and ltg_g_pipe_join) */
    void* ev_dec1;
    void* ev_tail; /* hipEvent_t, timing disabled: ltg_g_pipe_join's join of tail_stream (NULL: no tail stream) */
    float* h1pre;       /* [n_rows][H]  exchange 1: partial encoder pre-activation, all-reduced in place */
    float* rowpart_all; /* [n_ranks][n_rows][5]  exchange 2: all-gather IN PLACE (this rank writes block `rank`, n_rows = the call's batch) */
    float* dh2;         /* [n_rows][H]  exchange 3: local dh2, all-reduced in place */
    int32_t flags;      /* LTG_PIPE_* measurement switches, 0 = the shipped schedule */
    uint32_t seq;       /* ordinal of this call on this pipe: the caller adds 1 before every ltg_g_step_sharded call (starting at 1) */
    uint32_t* sync;     /* [16] device words, zeroed once by the caller: the hand-overs listed above (words 0-2, 5-8; 3-4: ltg_g_pipe_probe).
                         * NULL: the hand-overs are events (ev_fork / ev_dec1), as with LTG_PIPE_EVENTS */
    ltg_stream tail_stream; /* optional THIRD stream (ABI v12; device-word mode only): the step's Adam tail -- W_p0, W_q1 and the biases; it needs
                             * the backward's dh1 only -- runs there beside the sparse W_q0 gradient, the next call's catch-up and enc-0
                             * (words 9 / 10); used with LTG_PIPE_TAIL_OWN only.  NULL: the tail is the step's last kernel on the caller's stream */
    /* catch-up ahead (ABI v13; all optional, see above) */
    int32_t* q0_mark;           /* [n_items] int32, zeroed by the caller (and again whenever seq restarts): ordinal of the last call whose batch holds the row */
    const int32_t* next_uitem;  /* the NEXT call's ltg_batch.uitem (its distinct local items) or NULL */
    int32_t next_nu;            /* ... and its n_unique */
    int32_t caught_up;          /* 1: the previous call on this pipe was given THIS batch as next_uitem / next_nu and gen->q0_ord moved by that call only */
    /* second shadow buffer (ABI v13; optional, device-word mode): [n_items][608] bf16 like gen->wp1t_bf16, K padding zero.  The weight update
     * of the call writes the new shadow rows HERE instead of in place, so it starts when dlogits is complete -- beside the dh2 product, the
     * last reader of gen->wp1t_bf16 -- instead of behind it.  AFTER the call the caller exchanges gen->wp1t_bf16 and shadow_out. */
    uint16_t* shadow_out;
} ltg_pipe;
#define LTG_PIPE_NO_DEC1_FORK 1  /* everything on the caller's stream, in program order */
#define LTG_PIPE_NO_SLICE_FORK 2 /* the lazy clock's slice on the caller's stream, at the end of its own step */
#define LTG_PIPE_SLICE_IN_TOUCH 32 /* with device words: the slice step t - 1 owes in the catch-up launch of call t (as with events) instead of on the
                                      side stream between the catch-up of call t and that of call t + 1 */
#define LTG_PIPE_EVENTS 16       /* fork and join of the weight update as hipEventRecord / hipStreamWaitEvent pairs instead of device words */
#define LTG_PIPE_WIDE_GRAD 8     /* the sparse W_q0 gradient in its column-blocked shape (three times the waves) although it runs beside the update */
#define LTG_PIPE_TAIL_OWN 128    /* the Adam tail on ltg_pipe.tail_stream (default: the step's last kernel on the caller's stream -- measured, see ltg_g_step_sharded) */
/* bits 8-16 of flags: persistent workgroups of the forked weight update (0 = the library's choice) */

int ltg_g_step_sharded_ok(const ltg_config* cfg, const ltg_gen_state* gen, int32_t n_rows);
/* What ltg_g_step_sharded will do with this batch and this pipe (the caller's bookkeeping depends on it), bits:
 *   LTG_PLAN_AHEAD   it brings the rows announced in next_uitem up to date and honours caught_up
 *   LTG_PLAN_SHADOW  its weight update writes pipe->shadow_out (the caller exchanges it with gen->wp1t_bf16 after the call)
 * 0 also when the call would be refused. */
#define LTG_PLAN_AHEAD 1
#define LTG_PLAN_SHADOW 2
int ltg_g_step_sharded_plan(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_pipe* pipe);
int ltg_g_step_sharded(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* batch,
                       const ltg_pairs* fake, const ltg_g_opts* opts, const ltg_gen_acts* acts, const ltg_comm* comm,
                       const ltg_pipe* pipe, float* loss_out, void* ws, size_t ws_bytes, ltg_stream stream);
/* 1: `stream` and pipe->side_stream (and pipe->tail_stream, and those two with each other) run concurrently (a waiter on one does not hold
 * up the other), as the device-word hand-over needs;
 * 0: they share a hardware queue (HIP maps streams onto a few of them) -- use another side stream, or LTG_PIPE_EVENTS; < 0: error.
 * Synchronises both streams; uses sync[3..4].  Call it once per (stream, side stream) pair before the first step. */
int ltg_g_pipe_probe(const ltg_pipe* pipe, ltg_stream stream);
/* `stream` waits for the forked pieces of the last ltg_g_step_sharded call (dec1_stream, q0_stream). */
int ltg_g_pipe_join(const ltg_pipe* pipe, ltg_stream stream);
/* (re)build gen->wp1t_bf16 from gen->p[3] (after initialisation or after loading weights). */
int ltg_refresh_shadow(const ltg_config* cfg, const ltg_gen_state* gen, ltg_stream stream);
/* lazy Adam clock of W_q0 (ltg_gen_state.q0_last): apply every deferred zero-gradient step, all rows up to gen->q0_ord */
int ltg_g_flush(const ltg_config* cfg, const ltg_gen_state* gen, ltg_stream stream);
/* (re)build the e4m3 operand shadows of the discriminator (ltg_disc_state.emb_fp8 ... w3t_fp8) from the fp32 tensors. */
int ltg_refresh_d_shadow(const ltg_config* cfg, const ltg_disc_state* disc, ltg_stream stream);

/* Ranking metrics on device: replaces pred[X.nonzero()] = -inf (Codes/train.py:341) +
 * NDCG_binary_at_k_batch / Recall_at_k_batch (Codes/eval_functions.py:11-62).
 * tr = fold-in rows (masked), te = held-out rows, same n_rows.  out [n_rows][4] =
 * {ndcg@k_ndcg, recall@k_r1, recall@k_r2, valid flag (IDCG != 0)}. */
int ltg_rank_metrics(const ltg_config* cfg, const float* logits, const ltg_batch* tr,
                     const ltg_batch* te, int32_t k_ndcg, int32_t k_r1, int32_t k_r2, float* out,
                     ltg_stream stream);

/* The same metrics cut at their two exchange points for item-sharded runs (cfg->item_lo / n_items = this rank's
 * slab; tr holds LOCAL column ids of the slab, te GLOBAL ids; score/count arrays are indexed like te->indices, i.e.
 * by the absolute entry positions te->indptr holds):
 *   ltg_rank_scores  score_out[e] = logit of held-out entry e if this rank owns the item (-inf if it is a fold-in
 *                    item of that row, train.py:341), 0 otherwise                       -> all-reduce(sum)
 *   ltg_rank_counts  count_out[e] = #{local items ranked before entry e} (ties: lower GLOBAL id first, like
 *                    ltg_rank_metrics)                                                  -> all-reduce(sum)
 *   ltg_rank_finish  NDCG / Recall per row from the summed counts (eval_functions.py:24-38, :54-62). */
int ltg_rank_scores(const ltg_config* cfg, const float* logits, const ltg_batch* tr, const ltg_batch* te,
                    float* score_out, ltg_stream stream);
int ltg_rank_counts(const ltg_config* cfg, const float* logits, const ltg_batch* tr, const ltg_batch* te,
                    const float* score, int32_t* count_out, ltg_stream stream);
int ltg_rank_finish(const ltg_batch* te, const int32_t* counts, int32_t k_ndcg, int32_t k_r1, int32_t k_r2,
                    float* out, ltg_stream stream);

/* Verification helper of the LTG_PREC_FP8 mode: out[i] = the value the fp8 GEMM operands carry for in[i]
 * (clamp to +-448, round to nearest-even OCP e4m3) -- lets a test pin its CPU model of the rounding to the hardware. */
int ltg_fp8_roundtrip(const float* in, float* out, int32_t n, ltg_stream stream);
/* verification helper (ABI v14): the three bf16 terms (hi, mid, lo as fp32 values) into which the d_arith = BF16X6 loaders split every fp32 operand:
 * out[3 i + t] for in[i], n % 4 == 0.  The claim the tests pin: hi + mid + lo == in exactly, each term the round-to-nearest bf16 of its residual. */
int ltg_debug_split(const float* in, float* out, int32_t n, ltg_stream stream);
/* Verification helper: C[M][N] = A[M][K] . B[K][N] (row-major fp32) through the MFMA block template every GEMM-shaped
 * kernel is an instance of; mode 0 = fp32 operands, 1 = bf16, 2 = e4m3 with scale 2^4 on both operands. */
int ltg_debug_gemm(int32_t mode, int32_t M, int32_t N, int32_t K, const float* A, const float* B, float* C, ltg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* LTG_H */
