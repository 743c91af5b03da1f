// Part of csrc/ltg_fast.h (included there, in this order, inside ltg_kernels.hip's anonymous namespace): latency-path kernels of the generator: middle layers (fk_enc1, fk_dec0, fk_dz, fk_dh1) and encoder layer 0 (fk_enc0_fwd, fk_enc0_grad*).
// Split out of the 2 100-line header in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// generator middle layers (MultiVAE.py:152-172)
// ---------------------------------------------------------------------------------------------------------------------

// enc-1 + reparameterisation (MultiVAE.py:152,157-162, :178-181): mulv = h1 . W_q1 + b_q1 [B][2Z] and
// z = mu + is_training * eps * exp(logvar / 2) [B][Z].  A workgroup owns 16 rows x (16 columns of mu AND the same 16 columns
// of logvar): logical tile column c < 16 is column n0 + c, c >= 16 is column Z + n0 + (c - 16), so the epilogue holds both
// halves of a z value in two lanes 16 apart -- z is computed once per element, eps is drawn once per element.
typedef LtgRg<1, 2, 1, 1, 4> Rg16x32;   // 16 x 32 tile, four K slices
typedef LtgRg<1, 2, 1, 1, 8> Rg16x32k8; // the same tile over EIGHT K slices (512 threads): fk_enc1
// a value and what its transform needs, requested together (operand loaders of ltg_rgemm return it RAW; the a_xf functor folds it)
struct LtgRaw2 {
    ltg_f32x4 x, y;
};
// PRE (item-sharded step): h1 holds the all-reduced PRE-activation of enc-0; the operand loader applies bias + tanh
// (MultiVAE.py:152-155) and the column-tile-0 workgroups leave h1 = tanh(pre + b_q0) in h1_out for the backward -- no separate
// k_bias_tanh launch between the exchange and this layer.
// Round 5: EIGHT K slices (512 threads).  With four, a wave had 90 requests to issue -- 80 of them the strided 4-byte loads of the [K][N] weight
// operand -- and can have 64 in flight: the last third waited for the first arrivals (2.6 us until all were issued, profiles/r5_stamp_fk_enc1_fk_dh1.txt);
// with eight, 45 requests and 40 MFMAs per wave.
constexpr int ENC1_NT = 512;
template <bool PRE>
__global__ __launch_bounds__(ENC1_NT) void fk_enc1(int B, int H, int Z, const float* __restrict__ h1, const float* __restrict__ Wq1,
                                              const float* __restrict__ bq1, const float* __restrict__ eps_in, float is_training,
                                              uint64_t seed, uint64_t step, float* __restrict__ mulv, float* __restrict__ z,
                                              const float* __restrict__ bq0 = nullptr, float* __restrict__ h1_out = nullptr,
                                              LtgGate end_wait = LTG_NO_GATE) {
    // end_wait (one-call step): the kernel behind this one, dec-0, overwrites h2, which the previous step's weight update reads in its
    // prologue on the side stream -- ONE thread of this launch polls for the update's word as the last thing it does
    LTG_STAMP_AT(11, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg16x32k8::LDS_FLOATS];
    const LtgTile2 tl = xcd_tile2();
    const int m0 = tl.y * 16, n0 = tl.x * 16, Z2 = 2 * Z;
    auto col = [=] __device__(int c) { return min(n0 + (c & 15), Z - 1) + (c >> 4) * Z; };   // logical tile column -> column of mulv
    auto a_ld = [=] __device__(int, int m, int k) {
        if constexpr (PRE) return LtgRaw2{ltg_ld4(h1 + (size_t)m * H + k), ltg_ld4(bq0 + k)};
        else return ltg_ld4(h1 + (size_t)m * H + k);
    };
    const bool keep_h1 = PRE && tl.x == 0;
    auto a_xf = [=] __device__(auto raw, int, int m, int k) {
        if constexpr (PRE) {
            const ltg_f32x4 t{tanhf(raw.x[0] + raw.y[0]), tanhf(raw.x[1] + raw.y[1]), tanhf(raw.x[2] + raw.y[2]), tanhf(raw.x[3] + raw.y[3])};
            if (keep_h1) *reinterpret_cast<ltg_f32x4*>(h1_out + (size_t)m * H + k) = t;   // (clamped duplicates store the same value)
            return t;
        } else return raw;
    };
    auto b_ld = [=] __device__(int, int k, int c) { return ltg_ld4s(Wq1 + (size_t)k * Z2 + col(c), Z2); };
    // the epilogue's own operands are requested BEFORE the product (thread -> output map of ltg_rgemm: id = tid + 256 e,
    // row id / 32, logical column id % 32 = tid % 32), so the epilogue adds no round trip
    const float biasv = bq1[col(threadIdx.x & 31)];
    float epsv[1] = {0.f};          // (512 threads: one output of the 16 x 32 tile per thread)
    if (is_training != 0.f && eps_in)   // uniform
        epsv[0] = eps_in[(size_t)min(m0 + (int)threadIdx.x / 32, B - 1) * Z + min(n0 + (int)(threadIdx.x & 15), Z - 1)];
    auto epi = [=] __device__(int ei, int m, int c, float v, bool) {
        const int j = n0 + (c & 15);
        const bool ok = m < B && j < Z;
        const bool islv = c >= 16;
        const float mine = v + biasv;
        const float other = __shfl_xor(mine, 16);      // mu <-> logvar of the same z column
        if (ok) mulv[(size_t)m * Z2 + (islv ? Z : 0) + j] = mine;
        if (ok && !islv) {
            float e = 0.f;
            if (is_training != 0.f)   // uniform
                e = eps_in ? epsv[ei] : ltg_rng_normal(seed, LTG_STREAM_VAE_EPS, step, (uint64_t)m * Z + j);
            z[(size_t)m * Z + j] = mine + is_training * e * expf(0.5f * other);
        }
    };
    // the block sees logical columns [0, 32) of this tile (all "in range"; the real bounds are the functors' business)
#ifndef LTG_ENC1_SPL
#define LTG_ENC1_SPL 0      // (6: this product as six bf16 cross terms, ltg_rgemm.h -- measured in round 6 and NOT kept: a 1 x 2 tile per wave over eight K slices, the
                            // split's vector instructions cost what the matrix pipe gives back: G phase 63.27 -> 63.42 ms, profiles/r6_ab_generator_split.txt)
#endif
    ltg_rgemm<1, 2, 1, 1, 8, 5, false, false, 11, LTG_ENC1_SPL>(B, 32, H, m0, 0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds);      // (5 blocks of 16 per slice: H <= 640)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ltg_gate_wait_tail(end_wait);
}

// dec-0 (MultiVAE.py:168-172): h2 = tanh(z . W_p0 + b_p0); the column-tile-0 workgroups also add up the per-row KL
// (MultiVAE.py:161) from mulv.
__global__ __launch_bounds__(NT) void fk_dec0(int B, int H, int Z, const float* __restrict__ z, const float* __restrict__ mulv,
                                              const float* __restrict__ Wp0, const float* __restrict__ bp0, float* __restrict__ kl_rows,
                                              float* __restrict__ h2, LtgGate end_wait = LTG_NO_GATE, const unsigned* __restrict__ poison = nullptr) {
    // one-call step: h2 may only be overwritten once the previous step's weight update (side stream) has read it -- the kernel in front
    // of this one waited for that (fk_enc1's end_wait; poison: the wait gave up).  end_wait: the kernel BEHIND this one streams the bf16
    // shadow of W_p1t, which that update rewrites until it ends: one thread of this launch polls for its end as the last thing it does.
    LTG_STAMP_AT(12, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg16::LDS_FLOATS];
    const unsigned dead = ltg_poison_word(poison);   // (requested first, looked at in front of the stores: not a round trip of its own)
    const LtgTile2 tl = xcd_tile2();
    const int m0 = tl.y * 16, n0 = tl.x * 16;
    float kl = 0.f;
    if (tl.x == 0) {   // uniform: 16 threads per row, each a strided share of the row's Z columns
        const int rr = threadIdx.x >> 4, cc = threadIdx.x & 15;
        const float* mrow = mulv + (size_t)min(m0 + rr, B - 1) * 2 * Z;
        // Round 5: the thread's Z / 16 (mu, logvar) pairs requested AT ONCE (clamped, masked), added in the same order.  As a plain loop with a
        // runtime bound every pair was a round trip of its own (load, wait, add: 13 dependent trips at Z = 200), in the seven workgroups the
        // whole launch then waited for: 7.6 us for a 24-MFLOP product.
        constexpr int KLU = 16;
        float kmu[KLU], klv[KLU];
#pragma unroll
        for (int u = 0; u < KLU; ++u) {
            const int j = min(cc + 16 * u, Z - 1);
            kmu[u] = mrow[j];
            klv[u] = mrow[Z + j];
        }
#pragma unroll
        for (int u = 0; u < KLU; ++u)
            if (cc + 16 * u < Z) kl += 0.5f * (-klv[u] + expf(klv[u]) + kmu[u] * kmu[u] - 1.f);
        for (int j = cc + 16 * KLU; j < Z; j += 16) {      // (z_dim > 256)
            const float mu = mrow[j], lv = mrow[Z + j];
            kl += 0.5f * (-lv + expf(lv) + mu * mu - 1.f);
        }
    }
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(z + (size_t)m * Z + k); };
    auto b_ld = [=] __device__(int, int k, int n) { return ltg_ld4s(Wp0 + (size_t)k * H + n, H); };
    const float biasv = bp0[min(n0 + (int)(threadIdx.x & 15), H - 1)];
    auto epi = [=] __device__(int, int m, int n, float v, bool ok) {
        if (ok && !ltg_word_set(dead)) h2[(size_t)m * H + n] = tanhf(v + biasv);
    };
    ltg_rgemm<1, 1, 1, 1, 4, 4, false, false, 12>(B, H, Z, m0, n0, a_ld, LtgXfId(), b_ld, LtgXfId(), epi, lds);
    if (ltg_word_set(dead)) return;
    if (tl.x == 0) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) kl += __shfl_xor(kl, o);    // the 16 lanes of a row are consecutive
        if ((threadIdx.x & 15) == 0 && m0 + (threadIdx.x >> 4) < B) kl_rows[m0 + (threadIdx.x >> 4)] = kl;
    }
    if (tl.x == 1 && tl.y == 0 && threadIdx.x == 0) ltg_gate_wait_tail(end_wait);   // (tile column 1: not one that adds up the KL)
}

// dz = da2 . W_p0^T, then d mu / d logvar (KL + reparameterisation terms)          [B][2Z]
// DH2 (item-sharded step): `da2` holds the all-reduced dh2; the operand loader applies the tanh derivative with h2
// (da2 = dh2 (1 - h2^2), MultiVAE.py:168-172 backward) and the column-tile-0 workgroups leave da2 in da2_out for the weight
// gradients -- no separate k_da2 launch between the exchange and this layer.
template <bool DH2 = false>
__device__ __forceinline__ void dz_tile(int B, int Z, int H, const float* __restrict__ da2, const float* __restrict__ Wp0,
                                        const float* __restrict__ mulv, const float* __restrict__ eps_in, float is_training, float anneal,
                                        uint64_t seed, uint64_t step, float* __restrict__ dmlv, int m0, int n0, float* __restrict__ lds,
                                        const float* __restrict__ h2 = nullptr, float* __restrict__ da2_out = nullptr, bool keep = false) {
    const float invB = 1.f / (float)B;
    auto a_ld = [=] __device__(int, int m, int k) {
        if constexpr (DH2) return LtgRaw2{ltg_ld4(da2 + (size_t)m * H + k), ltg_ld4(h2 + (size_t)m * H + k)};
        else return ltg_ld4(da2 + (size_t)m * H + k);
    };
    auto a_xf = [=] __device__(auto raw, int, int m, int k) {
        if constexpr (DH2) {
            const ltg_f32x4 d{raw.x[0] * __builtin_fmaf(-raw.y[0], raw.y[0], 1.f), raw.x[1] * __builtin_fmaf(-raw.y[1], raw.y[1], 1.f),
                              raw.x[2] * __builtin_fmaf(-raw.y[2], raw.y[2], 1.f), raw.x[3] * __builtin_fmaf(-raw.y[3], raw.y[3], 1.f)};   // (k_da2's expression)
            if (keep) *reinterpret_cast<ltg_f32x4*>(da2_out + (size_t)m * H + k) = d;
            return d;
        } else return raw;
    };
    auto b_ld = [=] __device__(int, int k, int n) { return ltg_ld4(Wp0 + (size_t)n * H + k); };
    const int pm = min(m0 + (int)(threadIdx.x >> 4), B - 1), pn = min(n0 + (int)(threadIdx.x & 15), Z - 1);   // this thread's output
    const float mu = mulv[(size_t)pm * 2 * Z + pn], lv = mulv[(size_t)pm * 2 * Z + Z + pn];
    const float epsv = (is_training != 0.f && eps_in) ? eps_in[(size_t)pm * Z + pn] : 0.f;
    auto epi = [=] __device__(int, int m, int n, float dz, bool ok) {
        if (!ok) return;
        float e = 0.f;
        if (is_training != 0.f)
            e = eps_in ? epsv : ltg_rng_normal(seed, LTG_STREAM_VAE_EPS, step, (uint64_t)m * Z + n);
        dmlv[(size_t)m * 2 * Z + n] = dz + anneal * mu * invB;
        dmlv[(size_t)m * 2 * Z + Z + n] = dz * is_training * e * expf(0.5f * lv) * 0.5f + anneal * 0.5f * (expf(lv) - 1.f) * invB;
    };
    // (eight K slices, as in fk_enc1 / fk_dh2, measured here: G phase 66.29 -> 66.45 ms per epoch with fk_dh1 -- 20 requests per wave are no queue)
    ltg_rgemm<1, 1, 1, 1, 4, 10, false, false, 15>(B, Z, H, m0, n0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds);
}

// dh1 = dmlv . W_q1^T ; da1 = dh1 * (1 - h1^2)                                      [B][H]
__device__ __forceinline__ void dh1_tile(int B, int H, int Z2, const float* __restrict__ dmlv, const float* __restrict__ Wq1,
                                         const float* __restrict__ h1, float* __restrict__ da1, int m0, int n0, float* __restrict__ lds) {
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(dmlv + (size_t)m * Z2 + k); };
    auto b_ld = [=] __device__(int, int k, int n) { return ltg_ld4(Wq1 + (size_t)n * Z2 + k); };
    const float t = h1[(size_t)min(m0 + (int)(threadIdx.x >> 4), B - 1) * H + min(n0 + (int)(threadIdx.x & 15), H - 1)];
    auto epi = [=] __device__(int, int m, int n, float v, bool ok) {
        if (ok) da1[(size_t)m * H + n] = v * (1.f - t * t);
    };
    ltg_rgemm<1, 1, 1, 1, 4, 7, false, false, 16>(B, H, Z2, m0, n0, a_ld, LtgXfId(), b_ld, LtgXfId(), epi, lds);
}

__global__ __launch_bounds__(NT) void fk_dz(int B, int Z, int H, const float* __restrict__ da2, const float* __restrict__ Wp0,
                                            const float* __restrict__ mulv, const float* __restrict__ eps_in, float is_training,
                                            float anneal, uint64_t seed, uint64_t step, float* __restrict__ dmlv) {
    LTG_STAMP_AT(15, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg16::LDS_FLOATS];
    const LtgTile2 tl = xcd_tile2();
    dz_tile(B, Z, H, da2, Wp0, mulv, eps_in, is_training, anneal, seed, step, dmlv, tl.y * 16, tl.x * 16, lds);
}
__global__ __launch_bounds__(NT) void fk_dz_dh2(int B, int Z, int H, const float* __restrict__ dh2, const float* __restrict__ h2,
                                                const float* __restrict__ Wp0, const float* __restrict__ mulv, const float* __restrict__ eps_in,
                                                float is_training, float anneal, uint64_t seed, uint64_t step, float* __restrict__ dmlv,
                                                float* __restrict__ da2_out) {
    __shared__ __attribute__((aligned(16))) float lds[Rg16::LDS_FLOATS];
    const LtgTile2 tl = xcd_tile2();
    dz_tile<true>(B, Z, H, dh2, Wp0, mulv, eps_in, is_training, anneal, seed, step, dmlv, tl.y * 16, tl.x * 16, lds, h2, da2_out, tl.x == 0);
}
__global__ __launch_bounds__(NT) void fk_dh1(int B, int H, int Z2, const float* __restrict__ dmlv, const float* __restrict__ Wq1,
                                             const float* __restrict__ h1, float* __restrict__ da1) {
    LTG_STAMP_AT(16, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg16::LDS_FLOATS];
    const LtgTile2 tl = xcd_tile2();
    dh1_tile(B, H, Z2, dmlv, Wq1, h1, da1, tl.y * 16, tl.x * 16, lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// encoder layer 0: sparse row gather-sum and its sparse gradient (MultiVAE.py:148-155)
// ---------------------------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per (block of 256 columns, user row): a lane owns ONE float4 column chunk, so eight gathered
// W_q0 rows are in flight per wave and 128 per workgroup and trip -- a 900-item history takes 8 dependent trips instead of
// 15, a median row one.  (l2_normalize eps, dropout convention, item-shard conventions: see k_enc0_fwd.)
constexpr int E0_U = 8;
__global__ __launch_bounds__(ENC_NT) void fk_enc0_fwd(int H, int I, const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                      const float* __restrict__ values, const uint8_t* __restrict__ drop_keep, float keep,
                                                      uint64_t seed, uint64_t step, const float* __restrict__ Wq0,
                                                      const float* __restrict__ bq0, float* __restrict__ h1, float* __restrict__ row_scale,
                                                      const float* __restrict__ row_norm2, int item_lo, int Ig, int pre_only,
                                                      float* __restrict__ xd, int rps, LtgGate started = LTG_NO_GATE, LtgGate end_wait = LTG_NO_GATE) {
    // started (one-call step, slice on the side stream): opened as soon as this kernel runs -- the catch-up of the batch's rows in front
    // of it is complete.  end_wait (the Adam tail of the PREVIOUS call on its own stream): enc-1, behind this kernel, reads W_q1 and the
    // biases that tail updates and overwrites activations it reads -- one more block row (blockIdx.y == gridDim.y - 1) polls for its word
    if (end_wait.word && blockIdx.y == gridDim.y - 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ltg_gate_wait_tail(end_wait);
        return;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ltg_gate_set(started);
    // xd (optional, small item slabs): the dense row  xd[b][i] = keep_bi * x_bi / (keep * ||x_b||)  of the operand this
    // layer multiplies -- the backward forms dW_q0 = xd^T . da1 as a dense MFMA product with the very same dropout draw
    extern __shared__ __attribute__((aligned(16))) float s_row[];   // [I] when xd, else nothing
    __shared__ __attribute__((aligned(16))) float4 s_part[ENC_NW][64];
    __shared__ int s_idx[ENC_NT];
    __shared__ float s_val[ENC_NT];
    __shared__ float red[ENC_NW];
    const int b = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // several batches in one launch (ltg_fwd_opts.rows_per_step): the RNG sees the row's own batch counter and its row there
    const uint64_t kb = rps > 0 ? (uint64_t)(b % rps) : (uint64_t)b;
    step += rps > 0 ? (uint64_t)(b / rps) : 0;
    const int beg = indptr[b], end = indptr[b + 1];
    // (round 5: the bias chunk this thread adds at the very end is requested here -- behind the last barrier it was one more round trip)
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!pre_only && tid < 64 && 64 * cb + tid < (H >> 2)) bb = *reinterpret_cast<const float4*>(bq0 + 4 * (64 * cb + tid));
    float ss = 0.f;
    for (int e = beg + tid; e < end; e += ENC_NT) {
        const float v = values ? values[e] : 1.f;
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red[w] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < ENC_NW; ++i) ss += red[i];
    if (row_norm2) ss = row_norm2[b];
    const float scale = 1.f / (keep * sqrtf(fmaxf(ss, 1e-12f)));
    if (tid == 0 && cb == 0) row_scale[b] = scale;
    const int H4 = H >> 2;
    const int c4 = min(64 * cb + lane, H4 - 1);
    const bool dense = xd != nullptr && cb == 0;   // uniform
    if (dense)
        for (int i = tid; i < I; i += ENC_NT) s_row[i] = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = beg; c0 < end; c0 += ENC_NT) {
        __syncthreads();
        const int e = c0 + tid;
        if (e < end) {
            const int it = indices[e];
            const float v = values ? values[e] : 1.f;
            const bool kp = drop_keep ? (drop_keep[e] != 0)
                                      : ltg_rng_keep(seed, LTG_STREAM_VAE_DROPOUT, step, kb * (uint64_t)Ig + item_lo + it, keep);
            s_idx[tid] = it;
            s_val[tid] = kp ? v : 0.f;
            if (dense) s_row[it] = kp ? v * scale : 0.f;
        }
        __syncthreads();
        const int cnt = min(ENC_NT, end - c0);
        for (int j = w; j < cnt; j += E0_U * ENC_NW) {
            float v[E0_U];
            float4 x[E0_U];
#pragma unroll
            for (int u = 0; u < E0_U; ++u) {
                const int ju = j + u * ENC_NW;       // (wave-uniform: a slot beyond the chunk is skipped by a scalar branch, no request at all)
                v[u] = 0.f;
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                // (round 5: written as "row ju, or row j again with weight 0" the compiler loaded row j, WAITED for it, and requested the other
                // seven only under their masks with row j's value as the default -- two dependent trips per group instead of one; clamped to the
                // chunk's last entry instead, a 25-item row made 128 requests for the same row: +1.2 us per launch)
                if (ju < cnt) {
                    v[u] = s_val[ju];
                    x[u] = reinterpret_cast<const float4*>(Wq0 + (size_t)s_idx[ju] * H)[c4];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (all eight requests before the first sum: the scheduler otherwise holds the last one back
                                                    // behind the first two arrivals to save registers)
#pragma unroll
            for (int u = 0; u < E0_U; ++u) {
                acc.x += v[u] * x[u].x;
                acc.y += v[u] * x[u].y;
                acc.z += v[u] * x[u].z;
                acc.w += v[u] * x[u].w;
            }
        }
    }
    s_part[w][lane] = acc;
    __syncthreads();
    if (tid < 64 && 64 * cb + tid < H4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < ENC_NW; ++i) {
            const float4 p = s_part[i][tid];
            t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        const int c = 4 * (64 * cb + tid);
        float4 o;
        if (pre_only) o = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
        else {
            // (product ROUNDED before the bias is added -- no fma: it is the value an item-sharded run all-reduces, so that the one-call
            // sharded step, which applies bias + tanh in the next layer's loader, and this kernel yield the same bits)
            o = make_float4(tanhf(ltg_mul_rounded(t.x, scale) + bb.x), tanhf(ltg_mul_rounded(t.y, scale) + bb.y), tanhf(ltg_mul_rounded(t.z, scale) + bb.z),
                            tanhf(ltg_mul_rounded(t.w, scale) + bb.w));
        }
        *reinterpret_cast<float4*>(h1 + (size_t)b * H + c) = o;
    }
    if (dense) {   // (the barrier before the partial sums also ordered the scatter into s_row)
        float* dst = xd + (size_t)b * I;
        for (int i = 4 * tid; i < I; i += 4 * ENC_NT) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(s_row + i);   // I % 4 == 0
    }
}

// Sparse gradient rows of W_q0 (see k_enc0_grad): a 512-thread workgroup takes EIGHT gradient rows (distinct items of the
// batch, then the partial bias rows) of one block of 256 columns.  A row with at most G0_LIGHT entries -- almost every item of
// a large item slab occurs once or twice in a 100-user batch -- is summed by ONE wave (its entries all in flight at once); the
// head items of the popularity distribution (dozens of entries) are summed by the eight waves together, 8 entries in flight
// per wave, partials meeting in LDS.  4 500 one-item workgroups -> 570 at 20 000 items.  The eight rows of a workgroup are
// STRIDED over the row list (row j of group g = j * groups + g): the heavy rows are the lowest ids (popularity order) and
// would otherwise all sit in the first group and run one after the other (measured: 35 us instead of 14).
constexpr int G0_NT = 512, G0_NW = 8, G0_U = 8, G0_LIGHT = 16;
// entries q0, q0 + stride, ... < q1 of one gradient row, U in flight, over NCB chunks of 64 float4 columns (c4[k] = the lane's column in
// chunk k): acc[k] += scale * da1[row][c4[k]] in entry order, one fma per element (explicit: both shapes of the kernel must give the
// same bits, and the compiler's contraction choices differ from kernel to kernel)
template <int U, int NCB>
__device__ __forceinline__ void enc0_grad_entries(float4 (&acc)[NCB], int q0, int q1, int stride, bool is_item, const int (&c4)[NCB], int H4,
                                                  const int32_t* __restrict__ rowidx, const int32_t* __restrict__ csr_pos,
                                                  const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                  const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed, uint64_t step,
                                                  const float* __restrict__ row_scale, const float4* __restrict__ d4, int item_lo, int Ig, int item) {
    // item >= 0: the row's item is known (ltg_batch.uitem) -- with implicit values and the in-kernel dropout draw an entry then needs its
    // user row only: rowidx -> (row_scale, da1 row) instead of csr_pos -> indices -> ...
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool need_pos = item < 0 || values != nullptr || drop_keep != nullptr;   // uniform
    for (int q = q0; q < q1; q += U * stride) {
        int b[U];
        float sc[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int qt = q + t * stride;
            const bool ok = qt < q1;
            const int qc = ok ? qt : q;
            if (is_item) {   // uniform
                b[t] = rowidx[qc];
                const int pos = need_pos ? csr_pos[qc] : 0;
                const int it = item >= 0 ? item : indices[pos];
                const bool kp = drop_keep ? (drop_keep[pos] != 0)
                                          : ltg_rng_keep(seed, LTG_STREAM_VAE_DROPOUT, step, (uint64_t)b[t] * (uint64_t)Ig + item_lo + it, keep);
                sc[t] = (ok && kp) ? (values ? values[pos] : 1.f) * row_scale[b[t]] : 0.f;
            } else {
                b[t] = qc;
                sc[t] = ok ? 1.f : 0.f;
            }
        }
        float4 d[U][NCB];
#pragma unroll
        for (int t = 0; t < U; ++t)
#pragma unroll
            for (int k = 0; k < NCB; ++k) d[t][k] = d4[(size_t)b[t] * H4 + c4[k]];
#pragma unroll
        for (int t = 0; t < U; ++t)
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                acc[k].x = __builtin_fmaf(sc[t], d[t][k].x, acc[k].x);
                acc[k].y = __builtin_fmaf(sc[t], d[t][k].y, acc[k].y);
                acc[k].z = __builtin_fmaf(sc[t], d[t][k].z, acc[k].z);
                acc[k].w = __builtin_fmaf(sc[t], d[t][k].w, acc[k].w);
            }
    }
}
#ifndef LTG_G0_WAVES
#define LTG_G0_WAVES 6   // waves per SIMD the register allocation aims at (80 registers: three workgroups per CU; four registers spill)
#endif
__global__ __launch_bounds__(G0_NT, LTG_G0_WAVES) void fk_enc0_grad(int B, int I, int H, int nu, const int32_t* __restrict__ uptr,
                                                      const int32_t* __restrict__ rowidx, const int32_t* __restrict__ csr_pos,
                                                      const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                      const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed, uint64_t step,
                                                      const float* __restrict__ row_scale, const float* __restrict__ da1,
                                                      float* __restrict__ G, int item_lo, int Ig, ltg_gen_state st, AdamC ad, int lazy_ord,
                                                      const int32_t* __restrict__ uitem) {
    // lazy_ord > 0 (lazy Adam clock of W_q0, step `lazy_ord`): an item's gradient row is not stored -- the Adam step is applied to
    // its row of W_q0 / m / v right here (q0_touch brought every row of the batch to lazy_ord - 1 before the forward), the
    // workgroup of column block 0 moves the row's clock.  The partial bias rows still go to G (fk_g_tail sums them).
    // uitem (optional): the distinct items themselves.  The kernel is a chain of dependent round trips (7 without it: uptr -> rowidx,
    // csr_pos -> indices -> row_scale, da1 -> [uptr -> csr_pos -> indices ->] W / m / v); with it a light row takes 3 (uptr, uitem ->
    // rowidx + the row's W / m / v -> row_scale, da1) -- what the kernel costs beside the streaming weight update, where a round trip
    // queues behind ~20 MB of that kernel's requests.
    __shared__ __attribute__((aligned(16))) float4 s_g[G0_NW][64];
    const int cb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H4 = H >> 2, nrows = nu + ENC0_BIAS_PARTS;
    const int c4 = min(64 * cb + lane, H4 - 1);
    const int c4s[1] = {c4};
    const bool cok = 64 * cb + lane < H4;
    struct RowReq { float4 p, m, v; };
    auto row_off = [&](int i) { return (size_t)i * H4 + c4; };
    auto row_request = [&](int i) {
        return RowReq{reinterpret_cast<const float4*>(st.p[0])[row_off(i)], reinterpret_cast<const float4*>(st.m[0])[row_off(i)],
                      reinterpret_cast<const float4*>(st.v[0])[row_off(i)]};
    };
    auto adam_row = [&](int i, RowReq r, float4 g) {
        adam1(r.p.x, r.m.x, r.v.x, g.x, ad.lr_t, ad);
        adam1(r.p.y, r.m.y, r.v.y, g.y, ad.lr_t, ad);
        adam1(r.p.z, r.m.z, r.v.z, g.z, ad.lr_t, ad);
        adam1(r.p.w, r.m.w, r.v.w, g.w, ad.lr_t, ad);
        reinterpret_cast<float4*>(st.p[0])[row_off(i)] = r.p;
        reinterpret_cast<float4*>(st.m[0])[row_off(i)] = r.m;
        reinterpret_cast<float4*>(st.v[0])[row_off(i)] = r.v;
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) st.q0_last[i] = lazy_ord;
    };
    auto finish_row = [&](int u, float4 g, int item) {
        if (lazy_ord > 0 && u < nu) {
            const int i = item >= 0 ? item : indices[csr_pos[uptr[u]]];
            adam_row(i, row_request(i), g);
        } else {
            reinterpret_cast<float4*>(G)[(size_t)u * H4 + c4] = g;
        }
    };
    const float4* d4 = reinterpret_cast<const float4*>(da1);
    const int per = (B + ENC0_BIAS_PARTS - 1) / ENC0_BIAS_PARTS;
    // entry ranges of the group's eight rows (every wave computes all eight: the heavy / light split must be uniform)
    int q0[G0_NW], q1[G0_NW], uit[G0_NW];
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = min(j * (int)gridDim.y + (int)blockIdx.y, nrows - 1);
        const int bp = u - nu;
        q0[j] = u < nu ? uptr[u] : min(B, bp * per);
        q1[j] = u < nu ? uptr[u + 1] : min(B, (bp + 1) * per);
        uit[j] = (uitem && u < nu) ? uitem[u] : -1;
        if (j * (int)gridDim.y + (int)blockIdx.y >= nrows) q1[j] = q0[j];     // beyond the last row: empty
    }
    // light rows: wave j alone
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = j * (int)gridDim.y + (int)blockIdx.y;
        if (j == w && u < nrows && q1[j] - q0[j] <= G0_LIGHT) {
            const bool pre = lazy_ord > 0 && u < nu && uit[j] >= 0;   // (wave-uniform) the row's W / m / v travel while the entries are gathered
            RowReq r{};
            if (pre) r = row_request(uit[j]);
            float4 acc[1];
            enc0_grad_entries<G0_U, 1>(acc, q0[j], q1[j], 1, u < nu, c4s, H4, rowidx, csr_pos, indices, values, drop_keep, keep, seed, step, row_scale, d4,
                                       item_lo, Ig, uit[j]);
            if (cok) {
                if (pre) adam_row(uit[j], r, acc[0]);
                else finish_row(u, acc[0], uit[j]);
            }
        }
    }
    // heavy rows: all eight waves, one row after the other
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = j * (int)gridDim.y + (int)blockIdx.y;
        if (u < nrows && q1[j] - q0[j] > G0_LIGHT) {     // uniform over the workgroup
            float4 acc[1];
            enc0_grad_entries<G0_U, 1>(acc, q0[j] + w, q1[j], G0_NW, u < nu, c4s, H4, rowidx, csr_pos, indices, values, drop_keep, keep, seed, step, row_scale,
                                       d4, item_lo, Ig, uit[j]);
            __syncthreads();
            s_g[w][lane] = acc[0];
            __syncthreads();
            if (w == 0 && cok) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < G0_NW; ++i) {
                    const float4 p = s_g[i][lane];
                    t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
                }
                finish_row(u, t, uit[j]);
            }
        }
    }
}

// The same gradient rows with ONE wave per row over ALL columns (NCB chunks of 64 float4; H <= 768): a third of the waves of the
// column-blocked kernel above.  The shape for the one-call step, where the kernel runs beside the streaming decoder weight update: that
// update holds 196 of the 256 CUs (one 8-wave workgroup of 224 registers and 106 KB of LDS each), and what a kernel of the chain costs
// beside it is mostly how many ROUNDS its waves take on the 60 CUs left (measured with the update replaced by a dummy of its footprint and
// no memory traffic: fk_enc0_grad 15.5 -> 35-38 us, with the real update 38-46 us) -- 2 200 waves there, ~750 here.
// Same bits as fk_enc0_grad (same entry order per column, the heavy rows' eight chains, explicit fmas).
template <int NCB>
__global__ __launch_bounds__(G0_NT) void fk_enc0_grad_rows(int B, int I, int H, int nu, const int32_t* __restrict__ uptr,
                                                           const int32_t* __restrict__ rowidx, const int32_t* __restrict__ csr_pos,
                                                           const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                           const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed, uint64_t step,
                                                           const float* __restrict__ row_scale, const float* __restrict__ da1,
                                                           float* __restrict__ G, int item_lo, int Ig, ltg_gen_state st, AdamC ad, int lazy_ord,
                                                           const int32_t* __restrict__ uitem, const unsigned* __restrict__ poison = nullptr,
                                                           LtgGate started = LTG_NO_GATE, LtgGate end_wait = LTG_NO_GATE, float* __restrict__ lr_slot = nullptr) {
    // one-call step with the Adam tail on its own stream (ltg_pipe.tail_stream): `started` opens when this kernel runs -- dh1, the kernel
    // in front of it, is complete, which is all the tail waits for; lr_slot: this step's learning rate goes into the clock's ring HERE (the
    // next call's catch-up reads it, and the tail that used to write it now runs beside that catch-up); end_wait: this is then the last
    // kernel of the step on the caller's stream -- one more block at the end of the grid polls for the clock slice's word (see fk_g_tail)
    const int NG = (int)gridDim.x - (end_wait.word ? 1 : 0);     // groups of gradient rows
    if (end_wait.word && (int)blockIdx.x == NG) {
        if (threadIdx.x == 0) ltg_gate_wait_tail(end_wait);
        return;
    }
    if (ltg_poisoned(poison)) return;   // (one-call step: a device-side wait of the pipe gave up -- the model is not touched)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (lr_slot) *lr_slot = ad.lr_t;
        ltg_gate_set(started);
    }
    constexpr int U = NCB == 1 ? 8 : (NCB == 2 ? 4 : 3);   // entries in flight per wave (x NCB float4 each)
    __shared__ __attribute__((aligned(16))) float4 s_g[G0_NW][NCB * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H4 = H >> 2, nrows = nu + ENC0_BIAS_PARTS;
    int c4[NCB];
    bool cok[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
        c4[k] = min(64 * k + lane, H4 - 1);
        cok[k] = 64 * k + lane < H4;
    }
    const float4* d4 = reinterpret_cast<const float4*>(da1);
    const int per = (B + ENC0_BIAS_PARTS - 1) / ENC0_BIAS_PARTS;
    auto finish = [&](int u, int item, int k, float4 g, bool have, float4 p, float4 mm, float4 vv) {
        if (lazy_ord > 0 && u < nu) {
            const int i = item >= 0 ? item : indices[csr_pos[uptr[u]]];
            const size_t off = (size_t)i * H4 + c4[k];
            if (!have) {
                p = reinterpret_cast<const float4*>(st.p[0])[off];
                mm = reinterpret_cast<const float4*>(st.m[0])[off];
                vv = reinterpret_cast<const float4*>(st.v[0])[off];
            }
            adam1(p.x, mm.x, vv.x, g.x, ad.lr_t, ad);
            adam1(p.y, mm.y, vv.y, g.y, ad.lr_t, ad);
            adam1(p.z, mm.z, vv.z, g.z, ad.lr_t, ad);
            adam1(p.w, mm.w, vv.w, g.w, ad.lr_t, ad);
            reinterpret_cast<float4*>(st.p[0])[off] = p;
            reinterpret_cast<float4*>(st.m[0])[off] = mm;
            reinterpret_cast<float4*>(st.v[0])[off] = vv;
            if (k == 0 && lane == 0) st.q0_last[i] = lazy_ord;
        } else {
            reinterpret_cast<float4*>(G)[(size_t)u * H4 + c4[k]] = g;
        }
    };
    // entry ranges of the group's eight rows, strided over the row list as in fk_enc0_grad (every wave computes all eight: the heavy /
    // light split must be uniform)
    int q0[G0_NW], q1[G0_NW], uit[G0_NW];
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = min(j * NG + (int)blockIdx.x, nrows - 1);
        const int bp = u - nu;
        q0[j] = u < nu ? uptr[u] : min(B, bp * per);
        q1[j] = u < nu ? uptr[u + 1] : min(B, (bp + 1) * per);
        uit[j] = (uitem && u < nu) ? uitem[u] : -1;
        if (j * NG + (int)blockIdx.x >= nrows) q1[j] = q0[j];     // beyond the last row: empty
    }
    // light rows: wave j alone, its W / m / v rows requested before the gather
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = j * NG + (int)blockIdx.x;
        if (j == w && u < nrows && q1[j] - q0[j] <= G0_LIGHT) {
            const bool pre = lazy_ord > 0 && u < nu && uit[j] >= 0;   // wave-uniform
            float4 rp[NCB], rm[NCB], rv[NCB];
#pragma unroll
            for (int k = 0; k < NCB; ++k) {
                rp[k] = rm[k] = rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pre) {
                    const size_t off = (size_t)uit[j] * H4 + c4[k];
                    rp[k] = reinterpret_cast<const float4*>(st.p[0])[off];
                    rm[k] = reinterpret_cast<const float4*>(st.m[0])[off];
                    rv[k] = reinterpret_cast<const float4*>(st.v[0])[off];
                }
            }
            float4 acc[NCB];
            enc0_grad_entries<U, NCB>(acc, q0[j], q1[j], 1, u < nu, c4, H4, rowidx, csr_pos, indices, values, drop_keep, keep, seed, step, row_scale, d4,
                                      item_lo, Ig, uit[j]);
#pragma unroll
            for (int k = 0; k < NCB; ++k)
                if (cok[k]) finish(u, uit[j], k, acc[k], pre, rp[k], rm[k], rv[k]);
        }
    }
    // heavy rows: all eight waves (chain w = entries w, w + 8, ...), one row after the other; chunk k is finished by wave k
#pragma unroll
    for (int j = 0; j < G0_NW; ++j) {
        const int u = j * NG + (int)blockIdx.x;
        if (u < nrows && q1[j] - q0[j] > G0_LIGHT) {     // uniform over the workgroup
            float4 acc[NCB];
            enc0_grad_entries<U, NCB>(acc, q0[j] + w, q1[j], G0_NW, u < nu, c4, H4, rowidx, csr_pos, indices, values, drop_keep, keep, seed, step, row_scale,
                                      d4, item_lo, Ig, uit[j]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NCB; ++k) s_g[w][64 * k + lane] = acc[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NCB; ++k)
                if (w == k && cok[k]) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < G0_NW; ++i) {
                        const float4 p = s_g[i][64 * k + lane];
                        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
                    }
                    finish(u, uit[j], k, t, false, t, t, t);
                }
        }
    }
}
