// Latency-bound kernels of the path rebuilt on the register-resident MFMA block (ltg_rgemm.h): one memory round trip per
// workgroup instead of one per K tile.  Included by ltg_kernels.hip inside its anonymous namespace (after AdamC, PairView,
// DropView, DLayout, dact, Workspace, Probe).  Reference citations are relative to /root/reference/.
//
//   generator middle layers   fk_enc1, fk_dec0 (reparameterisation + KL folded into its operand loader), fk_dz, fk_dh1
//   encoder layer 0           fk_enc0_fwd / fk_enc0_grad: column-blocked, 8 gathered rows in flight per wave
//   discriminator             fk_d_l1, fk_d_l2 (output unit folded in: per-tile partial dot products), fk_d_y,
//                             fk_d_bwd1 / fk_d_bwd2 (gradient slabs), fk_d_adam (flat float4 sweep)
//   small item counts         fk_dec1, fk_row_dlogits (softmax statistics + losses + dlogits of a row in one pass),
//                             fk_dh2, fk_g_tail (the Adam updates of the generator as jobs of ONE launch)
#pragma once

#include "ltg_rgemm.h"

// Workgroups b, b + 8, b + 16 ... share an XCD (and its L2) under the observed round-robin placement: give each XCD a CONTIGUOUS
// run of the n tile ids, so that tiles of neighbouring rows (same A rows, all of B) meet in one L2 instead of all eight.
// Speed only: any placement computes the same result.  Bijective for every n.
__device__ __forceinline__ int xcd_chunk(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// The same for a 2-D grid of (column tile, row tile) workgroups: blockIdx -> tile ids with the ROW tiles of one column tile -- which share
// the column block of the weight matrix, the large operand of these latency-bound products -- consecutive, so that they meet in one XCD's L2.
// Dealt round-robin every row tile of a column fetched that block into another L2: fk_dec1 moved 21 MB per launch for 3 MB of operands,
// fk_enc1 10 MB for 1.2 (profiles/r4_pmc_traffic.json).  Round 5, Askubuntu_Sample G step with this order in fk_g_tail, fk_dec1, fk_dh2:
// 81.8 -> 78.9 ms per epoch on one box.
struct LtgTile2 {
    int x, y;   // column tile, row tile
};
__device__ __forceinline__ LtgTile2 xcd_tile2() {
    const int t = xcd_chunk((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    return LtgTile2{t / (int)gridDim.y, t % (int)gridDim.y};
}

// a * b rounded to fp32 on its own: never contracted into an fma with a following addition (HIP compiles with
// -ffp-contract=fast-honor-pragmas; __fmul_rn is a plain product there)
__device__ __forceinline__ float ltg_mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

typedef LtgRg<1, 1, 1, 1, 4> Rg16;    // 16 x 16 tile, four K slices
typedef LtgRg<2, 2, 1, 1, 4> Rg32k;   // 32 x 32 tile, four K slices (each wave the whole tile)
typedef LtgRg<2, 2, 1, 1, 8> Rg32k8;  // the same over EIGHT K slices (512 threads): fk_d_l2
typedef LtgRg<1, 1, 2, 2, 1> Rg32;    // 32 x 32 tile, one 16 x 16 per wave over the whole K

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
    ltg_rgemm<1, 2, 1, 1, 8, 5, false, false, 11>(B, 32, H, m0, 0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds);      // (5 blocks of 16 per slice: H <= 640)
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

// ---------------------------------------------------------------------------------------------------------------------
// discriminator (discriminator.py:3-58), fp32 operands, default-sized layers
// ---------------------------------------------------------------------------------------------------------------------

// branch layers (discriminator.py:16-19,25,30,51,52): blockIdx.z = 0 popular -> h1, 1 niche -> h2.  32 x 32 tiles, each wave
// a 16 x 16 product over the whole K = h0 (7 blocks in flight).
// SPL (all four GEMM kernels of the step): 0 = v_mfma_f32_16x16x4_f32, 6 / 4 = the bf16 cross terms of the split operands (ltg_rgemm.h; ltg_config.d_arith)
template <int SPL>
__global__ __launch_bounds__(NT) void fk_d_l1(PairView pv, int h0, int h1, int h2, const float* __restrict__ emb,
                                              const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                              const float* __restrict__ b2, DropView dA, DropView dB, float keep, uint64_t seed,
                                              uint64_t step, float* __restrict__ A1) {
#ifdef LTG_D_EMPTY   // MEASUREMENT BUILD ONLY (results wrong): the grid, registers and LDS of this launch, no work -- the D step's launch structure
    if (pv.nr >= 0) return;
#endif
    LTG_STAMP_AT(1, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg32::LDS_FLOATS];
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const bool br = blockIdx.z != 0;
    const int N = br ? h2 : h1;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;      // (round 5: the column tiles of a row tile in one XCD's L2 -- no difference: 57.2-57.5 us either way)
    if (n0 >= N) return;
    const float* W = br ? w2 : w1;
    const float* bias = br ? b2 : b1;
    const int coff = br ? h1 : 0;
    // the embedding row of this lane's operand row (one id load, not one per k block)
    const int myrow = min(Rg32::row(m0, 0), n - 1);
    const int id = br ? pv.nic(myrow) : pv.pop(myrow);
    const float* erow = emb + (size_t)max(id, 0) * h0;
    auto a_ld = [=] __device__(int, int, int k) { return ltg_ld4(erow + k); };
    auto a_xf = [=] __device__(ltg_f32x4 v, int, int, int) { return id >= 0 ? v : ltg_f32x4{0.f, 0.f, 0.f, 0.f}; };
    auto b_ld = [=] __device__(int, int k, int nn) { return ltg_ld4s(W + (size_t)k * N + nn, N); };
    const float biasv = bias[min(n0 + (int)(threadIdx.x & 31), N - 1)];
    auto epi = [=] __device__(int, int m, int nn, float v, bool ok) {
        if (!ok) return;
        const float t = tanhf(v + biasv);
        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep) : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
        A1[(size_t)m * h12 + coff + nn] = kp ? t / keep : 0.f;
    };
    ltg_rgemm<1, 1, 2, 2, 1, 7, false, true, 1, SPL>(n, N, h0, m0, n0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds);
}

// fully connected layer + the output unit's dot product (discriminator.py:44-45, :54-55): A3 = dropout(tanh(A1 . w3 + b3));
// G3 = w4 * d A3 / d pre (the factor the backward needs, so that dpre3 = ds[row] * G3); spart[tile_n][row] = this column
// tile's share of A3[row] . w4 -- the consumers add the tiles up in a fixed order (no atomics: reproducible).
// (round 5: eight K slices, 512 threads -- 35 instead of 70 requests and 56 instead of 112 MFMAs per wave; see fk_enc1)
constexpr int DL2_NT = 512;
template <int SPL>
__global__ __launch_bounds__(DL2_NT) void fk_d_l2(int n, int h12, int h3, const float* __restrict__ A1, const float* __restrict__ w3,
                                              const float* __restrict__ b3, const float* __restrict__ w4, DropView dC, float keep,
                                              uint64_t seed, uint64_t step, float* __restrict__ A3, float* __restrict__ G3,
                                              float* __restrict__ spart) {
#ifdef LTG_D_EMPTY   // MEASUREMENT BUILD ONLY (results wrong): the grid, registers and LDS of this launch, no work -- the D step's launch structure
    if (n >= 0) return;
#endif
    LTG_STAMP_AT(2, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg32k8::LDS_FLOATS];
    const int tn = (h3 + 31) / 32;
    const int tid_ = xcd_chunk(blockIdx.x, gridDim.x);
    const int m0 = (tid_ / tn) * 32, n0 = (tid_ % tn) * 32;
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(A1 + (size_t)m * h12 + k); };
    auto b_ld = [=] __device__(int, int k, int nn) { return ltg_ld4s(w3 + (size_t)k * h3 + nn, h3); };
    const int tile = tid_ % tn;
    const int pcol = min(n0 + (int)(threadIdx.x & 31), h3 - 1);
    const float b3v = b3[pcol], wv = w4[pcol];
    auto epi = [=] __device__(int, int m, int nn, float v, bool ok) {
        const int nc = min(nn, h3 - 1), mc = min(m, n - 1);
        const float t = tanhf(v + b3v);
        const bool kp = ok && dC.keep(mc, nc, h3, seed, LTG_STREAM_D_DROP_C, step, keep);
        const float a3 = kp ? t / keep : 0.f;
        if (ok) {
            A3[(size_t)m * h3 + nn] = a3;
            if (G3) G3[(size_t)m * h3 + nn] = wv * dact(a3, keep);
        }
        float pd = a3 * wv;   // the 32 columns of a tile row sit in 32 consecutive lanes
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) pd += __shfl_xor(pd, o);
        if ((threadIdx.x & 31) == 0 && m < n) spart[(size_t)tile * n + m] = pd;
    };
    ltg_rgemm<2, 2, 1, 1, 8, 4, false, true, 2, SPL>(n, h3, h12, m0, n0, a_ld, LtgXfId(), b_ld, LtgXfId(), epi, lds);
}

// output unit from the tile partials (discriminator.py:45,55; train.py:142): y, d loss / d s, loss term of one pair row
__device__ __forceinline__ void d_row_terms(const PairView& pv, int r, int n, int ntile, const float* __restrict__ spart, float b4v,
                                            float& yv, float& ds, float& lrow) {
    float s = b4v;
    for (int t = 0; t < ntile; ++t) s += spart[(size_t)t * n + r];
    const float yy = 1.f / (1.f + expf(-s));
    const bool ok = pv.valid(r), real = r < pv.nr;
    yv = ok ? yy : 0.f;
    ds = ok ? (real ? -(1.f - yy) : yy) : 0.f;
    lrow = ok ? (real ? -logf(yy) : -logf(1.f - yy)) : 0.f;
}

// y of every pair row (the generator step only needs sum_j y_j of the fake tower, train.py:155)
__global__ __launch_bounds__(NT) void fk_d_y(PairView pv, int ntile, const float* __restrict__ spart, const float* __restrict__ b4,
                                             float* __restrict__ y) {
    const int n = pv.nr + pv.nf;
    const int r = blockIdx.x * NT + threadIdx.x;
    if (r >= n) return;
    float yv, ds, lr;
    d_row_terms(pv, r, n, ntile, spart, b4[0], yv, ds, lr);
    y[r] = yv;
}

#ifndef LTG_BWD1_NA
#define LTG_BWD1_NA 5      // 16-deep k blocks a wave of job A / job B keeps in flight per pass (registers: 16 per block).  Measured with
                           // 3 / 2 instead (two passes, 138 -> ~85 registers, twice the resident workgroups): D step 59.5-60.5 us either way
#endif
#ifndef LTG_BWD1_NB
#define LTG_BWD1_NB 4
#endif
// Backward stage 1, ONE launch, three jobs by block index (gradient slabs are summed by the Adam sweep):
//   job A  dpre1 = ((ds G3) . w3^T) * dact(A1)                               [n][h1+h2]   32 x 32 tiles
//   job B  slab[z] = A1^T . (ds G3) (+ ones row -> db3), split over row chunks [h12+1][h3]  32 x 32 tiles
//   job C  slab[z]: dw4 = A3^T . ds, db4 = sum ds, and the chunk's share of d_loss (slot P of the slab)
// Every job first rebuilds ds (and the loss terms) of the pair rows it touches from the tile partials of fk_d_l2.
template <int SPL>
__device__ __forceinline__ void d_bwd1_jobs_bc(const PairView& pv, int h12, int h3, int bid, int nB, int ntile, const DLayout& L, int SP,
                                               const float* __restrict__ A1, const float* __restrict__ A3, const float* __restrict__ G3,
                                               const float* __restrict__ spart, float b4v, float* __restrict__ slab, float* __restrict__ lds,
                                               float* __restrict__ s_ds, float* __restrict__ s_lr);
template <int SPL>
__global__ __launch_bounds__(NT) void fk_d_bwd1(PairView pv, int h12, int h3, int nA, int nB, int ntile, DLayout L, int SP,
                                                const float* __restrict__ A1, const float* __restrict__ A3, const float* __restrict__ G3,
                                                const float* __restrict__ spart, const float* __restrict__ b4p,
                                                const float* __restrict__ w3, float keep, float* __restrict__ dpre1,
                                                float* __restrict__ slab, LtgGate started = LTG_NO_GATE) {
    // started (the step's jobs B / C on the caller's aux stream, ltg_d_opts.aux_stream): opened when this launch -- job A alone then --
    // runs: the forward in front of it is complete, which is all jobs B / C wait for
    if (blockIdx.x == 0 && threadIdx.x == 0) ltg_gate_set(started);
#ifdef LTG_D_EMPTY   // MEASUREMENT BUILD ONLY (results wrong): the grid, registers and LDS of this launch, no work -- the D step's launch structure
    if (pv.nr >= 0) return;
#endif
    LTG_STAMP_AT(3, 0);
    LTG_STAMP_AT(4, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg32k::LDS_FLOATS];
    __shared__ float s_ds[D_KCHUNK], s_lr[D_KCHUNK];
    const int n = pv.nr + pv.nf, tid = threadIdx.x;
    const float b4v = b4p[0];
    int bid = blockIdx.x;
    if (bid < nA) {
        const int tn = (h12 + 31) / 32;
        const int nAt = ((n + 31) / 32) * tn;          // real job-A tiles; nA is padded to a multiple of 8 (job B starts on XCD 0)
        const int ta = xcd_chunk(bid, nA);
        if (ta >= nAt) return;
        const int m0 = (ta / tn) * 32, n0 = (ta % tn) * 32;
        // requested up front, consumed later: the tile partials of this thread's pair row (threads 0..31) and the A1 values of
        // the four outputs this thread finishes -- neither costs a round trip of its own
        const int prow = min(m0 + (tid & 31), n - 1), pt = tid >> 5;            // thread -> (pair row, tile pt and pt + 8)
        const float sp0 = spart[(size_t)min(pt, ntile - 1) * n + prow], sp1 = spart[(size_t)min(pt + 8, ntile - 1) * n + prow];
        float a1v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = tid + 256 * e;
            a1v[e] = A1[(size_t)min(m0 + id / 32, n - 1) * h12 + min(n0 + id % 32, h12 - 1)];
        }
        const bool pvalid = pv.valid(prow), preal = prow < pv.nr;
        auto mid = [=] __device__() {
            s_lr[tid] = (pt < ntile ? sp0 : 0.f) + (pt + 8 < ntile ? sp1 : 0.f);       // ntile <= 16 (d_fast)
            __syncthreads();
            if (tid < 32) {
                float sv = b4v;
#pragma unroll
                for (int t = 0; t < 8; ++t) sv += s_lr[32 * t + tid];
                const float yy = 1.f / (1.f + expf(-sv));
                s_ds[tid] = pvalid ? (preal ? -(1.f - yy) : yy) : 0.f;
            }
            __syncthreads();
        };
        auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(G3 + (size_t)m * h3 + k); };
        auto a_xf = [=] __device__(ltg_f32x4 v, int, int m, int) { return v * s_ds[m - m0]; };
        auto b_ld = [=] __device__(int, int k, int nn) { return ltg_ld4(w3 + (size_t)nn * h3 + k); };
        auto epi = [=] __device__(int e, int m, int nn, float v, bool ok) {
            if (ok) dpre1[(size_t)m * h12 + nn] = v * dact(a1v[e], keep);
        };
        ltg_rgemm<2, 2, 1, 1, 4, LTG_BWD1_NA, false, true, 3, SPL>(n, h12, h3, m0, n0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds, mid);
        return;
    }
    d_bwd1_jobs_bc<SPL>(pv, h12, h3, bid - nA, nB, ntile, L, SP, A1, A3, G3, spart, b4v, slab, lds, s_ds, s_lr);
}

// jobs B and C of backward stage 1 (see fk_d_bwd1) for block `bid` of nB + nC: they need the forward's outputs only, not dpre1 --
// either kernel of the backward may carry them (d_step_impl: beside job A, or beside the embedding products of stage 2)
template <int SPL>
__device__ __forceinline__ void d_bwd1_jobs_bc(const PairView& pv, int h12, int h3, int bid, int nB, int ntile, const DLayout& L, int SP,
                                               const float* __restrict__ A1, const float* __restrict__ A3, const float* __restrict__ G3,
                                               const float* __restrict__ spart, float b4v, float* __restrict__ slab, float* __restrict__ lds,
                                               float* __restrict__ s_ds, float* __restrict__ s_lr) {
    const int n = pv.nr + pv.nf, tid = threadIdx.x;
    if (bid < nB) bid = xcd_chunk(bid, nB);            // a chunk of job B = the tiles of one or two row chunks z
    const int tmB = (h12 + 1 + 31) / 32, tnB = (h3 + 31) / 32;
    const int z = bid < nB ? bid / (tmB * tnB) : (bid - nB) / ((h3 + 2 + 31) / 32);
    const int kbeg = z * D_KCHUNK, kend = min(n, kbeg + D_KCHUNK), K = kend - kbeg;
    // ds / loss term of pair row kbeg + tid: the tile partials are requested here and consumed in the product's mid hook
    // (job B) -- after the operand requests have been issued, so the prologue costs no round trip of its own
    const int prow = min(kbeg + tid, n - 1);
    float sp[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) sp[t] = spart[(size_t)min(t, ntile - 1) * n + prow];
    const bool pvalid = pv.valid(prow) && kbeg + tid < kend, preal = prow < pv.nr;
    auto rows_to_lds = [=] __device__() {
        float sv = b4v;
#pragma unroll
        for (int t = 0; t < 16; ++t) sv += t < ntile ? sp[t] : 0.f;     // ntile <= 16 (d_fast)
        const float yy = 1.f / (1.f + expf(-sv));
        s_ds[tid] = pvalid ? (preal ? -(1.f - yy) : yy) : 0.f;
        s_lr[tid] = pvalid ? (preal ? -logf(yy) : -logf(1.f - yy)) : 0.f;
        __syncthreads();
    };
    float* out = slab + (size_t)z * SP;
    if (bid < nB) {
        const int t = bid % (tmB * tnB);
        const int m0 = (t / tnB) * 32, n0 = (t % tnB) * 32;
        const int ow = L.off[4], ob = L.off[5];
        auto a_ld = [=] __device__(int, int m, int k) {
            ltg_f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = A1[(size_t)(kbeg + min(k + j, K - 1)) * h12 + min(m, h12 - 1)];
            return v;
        };
        auto a_xf = [=] __device__(ltg_f32x4 x, int, int m, int k) {
            ltg_f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = k + j < K ? (m < h12 ? x[j] : 1.f) : 0.f;
            return v;
        };
        auto b_ld = [=] __device__(int, int k, int nn) {
            ltg_f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = G3[(size_t)(kbeg + min(k + j, K - 1)) * h3 + nn];
            return v;
        };
        auto b_xf = [=] __device__(ltg_f32x4 x, int, int k, int) {
            ltg_f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = s_ds[min(k + j, K - 1)] * x[j];
            return v;
        };
        auto epi = [=] __device__(int, int m, int nn, float g, bool ok) {
            if (!ok) return;
            if (m < h12) out[ow + (size_t)m * h3 + nn] = g;
            else out[ob + nn] = g;
        };
        ltg_rgemm<2, 2, 1, 1, 4, LTG_BWD1_NB, false, true, 4, SPL>(h12 + 1, h3, K, m0, n0, a_ld, a_xf, b_ld, b_xf, epi, lds, rows_to_lds);
        return;
    }
    bid -= nB;
    rows_to_lds();
    {
        // columns c < h3: dw4[c]; c == h3: db4; c == h3 + 1: the chunk's loss sum
        float (*part)[33] = reinterpret_cast<float (*)[33]>(lds);
        const int tc = (h3 + 2 + 31) / 32;
        const int tn = tid & 31, tr = tid >> 5;
        const int c = (bid % tc) * 32 + tn;
        float acc = 0.f;
        if (c <= h3 + 1) {
#pragma unroll 8
            for (int r = tr; r < K; r += 8) acc += (c < h3 ? A3[(size_t)(kbeg + r) * h3 + c] : 1.f) * (c == h3 + 1 ? s_lr[r] : s_ds[r]);
        }
        part[tr][tn] = acc;
        __syncthreads();
        if (tr == 0 && c <= h3 + 1) {
            float g = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) g += part[i][tn];
            out[c < h3 ? L.off[6] + c : (c == h3 ? L.off[7] : L.off[8])] = g;
        }
    }
}

// Backward stage 2: dw1 / db1 and dw2 / db2 slabs (E_pop^T . dpre1[:, :h1], E_niche^T . dpre1[:, h1:]), 16 x 32 tiles.
// (round 5: eight K slices, 512 threads -- two 16-deep blocks of a 256-row chunk per wave: 8 ids + 24 operand requests instead of 16 + 48; see fk_enc1)
constexpr int DB2_NT = 512;
typedef LtgRg<1, 2, 1, 1, 8> Rg16x32k8b;
template <int SPL>
__global__ __launch_bounds__(DB2_NT) void fk_d_bwd2(PairView pv, int h0, int h1, int h2, DLayout L, int SP, const float* __restrict__ emb,
                                                const float* __restrict__ dpre1, float* __restrict__ slab, LtgGate end_wait = LTG_NO_GATE) {
    // end_wait (jobs B / C of stage 1 on the aux stream): the Adam sweep behind this kernel adds THEIR slab entries too -- one more block
    // at the end of the grid polls for their word
    if (end_wait.word && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 0) ltg_gate_wait_tail(end_wait);
        return;
    }
#ifdef LTG_D_EMPTY   // MEASUREMENT BUILD ONLY (results wrong): the grid, registers and LDS of this launch, no work -- the D step's launch structure
    if (pv.nr >= 0) return;
#endif
    LTG_STAMP_AT(5, 0);
    __shared__ __attribute__((aligned(16))) float lds[Rg16x32k8b::LDS_FLOATS];
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const int tm = (h0 + 1 + 15) / 16;
    const int tn1 = (h1 + 31) / 32, tn2 = (h2 + 31) / 32;
    const int per_z = tm * (tn1 + tn2);
    // (Round 5, measured and removed: the tiles of one row chunk z in contiguous runs per XCD -- D step 57.7-58.2 against 57.2-57.4 us.)
    const int z = blockIdx.x / per_z, t = blockIdx.x % per_z;
    const int m0 = (t / (tn1 + tn2)) * 16;
    const int tcol = t % (tn1 + tn2);
    const bool br = tcol >= tn1;
    const int n0 = (br ? tcol - tn1 : tcol) * 32;
    const int N = br ? h2 : h1;
    const int coff = br ? h1 : 0;
    const int ow = L.off[br ? 2 : 0], ob = L.off[br ? 3 : 1];
    const int kbeg = z * D_KCHUNK, kend = min(n, kbeg + D_KCHUNK), K = kend - kbeg;
    float* out = slab + (size_t)z * SP;
    // phase 0: the pair ids of the 8 pair rows this lane multiplies (2 blocks x 4): their embedding rows are the dependent
    // second round trip
    constexpr int DB2_NB = D_KCHUNK / 16 / 8;      // 16-deep blocks per K slice
    int ids[DB2_NB][4];
#pragma unroll
    for (int i = 0; i < DB2_NB; ++i) {
        const int kc = Rg16x32k8b::kc(K, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = kbeg + min(kc + j, K - 1);
            ids[i][j] = br ? pv.nic(row) : pv.pop(row);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto a_ld = [=] __device__(int i, int m, int) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = emb[(size_t)max(ids[i][j], 0) * h0 + min(m, h0 - 1)];
        return v;
    };
    auto a_xf = [=] __device__(ltg_f32x4 x, int i, int m, int k) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = k + j < K ? (m < h0 ? (ids[i][j] >= 0 ? x[j] : 0.f) : 1.f) : 0.f;
        return v;
    };
    auto b_ld = [=] __device__(int, int k, int nn) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = dpre1[(size_t)(kbeg + min(k + j, K - 1)) * h12 + coff + nn];
        return v;
    };
    auto epi = [=] __device__(int, int m, int nn, float g, bool ok) {
        if (!ok) return;
        if (m < h0) out[ow + (size_t)m * N + nn] = g;
        else out[ob + nn] = g;
    };
    ltg_rgemm<1, 2, 1, 1, 8, DB2_NB, false, true, 5, SPL>(h0 + 1, N, K, m0, n0, a_ld, a_xf, b_ld, LtgXfId(), epi, lds);
}

// (Measured and not kept, round 3: this sweep FUSED into fk_d_bwd2 -- every workgroup releases its slab tile with a device-scope fence
// and takes a ticket on the tile's counter, the last arriver of a tile adds the chunk slabs and applies Adam, trailing blocks sweep
// w3 / b3 / w4 / b4.  Bit-identical and one launch fewer, but the ~1 000 release fences (an L2 write-back each) serialise: D step
// 59.8 -> 145 us on Askubuntu_Sample.  The launch boundary is the cheaper device-wide release.)
// One Adam sweep over the discriminator's trainable tensors laid out back to back (train.py:163): g = sum of the chunk
// slabs; 16 bytes per lane.  Block 0 also adds up d_loss (train.py:142) from slot P of the slabs.
__global__ __launch_bounds__(NT) void fk_d_adam(int ks, int P, int SP, const float* __restrict__ slab, float* __restrict__ p,
                                                float* __restrict__ m, float* __restrict__ v, AdamC ad, float* __restrict__ loss_out,
                                                const unsigned* __restrict__ poison = nullptr) {
    // (the poison word -- the wait for the aux stream's jobs gave up: the discriminator is not touched -- is REQUESTED first and looked at in
    // front of the first store: as the guard of an early return it was a round trip of its own in front of every other request of a launch
    // that is nothing but round trips)
    const unsigned dead = ltg_poison_word(poison);
#ifdef LTG_D_EMPTY   // MEASUREMENT BUILD ONLY (results wrong): the grid, registers and LDS of this launch, no work -- the D step's launch structure
    if (ks >= 0) return;
#endif
    const int P4 = P >> 2;
    constexpr int DA_U = 8;
    // Workgroup 0 (dispatched first) does the ragged tail and d_loss and nothing else; the sweep belongs to workgroups 1 .. gridDim.x - 1.
    // (Round 5: as the epilogue of workgroup 0's share of the sweep, these two serial walks were a second and a third chain of round trips
    // that the whole launch waited for.)
    // (The one launch site passes gridDim.x = sweep workgroups + 1.  A grid of ONE workgroup would leave nobody for the sweep: it then does the
    // sweep itself behind its side job -- uniform per launch, never taken by the library's own launch.)
    const bool alone = gridDim.x == 1;
    if (blockIdx.x == 0) {
        const int e = 4 * P4 + threadIdx.x;
        const bool tail = e < P, lossl = threadIdx.x == NT - 1;      // (the loss on another wave than the tail elements)
        const int col = tail ? e : P;
        if (!tail && !lossl && !alone) return;
        if (tail || lossl) {
        float pe = 0.f, me = 0.f, ve = 0.f;
        if (tail) { pe = p[e]; me = m[e]; ve = v[e]; }
        float t = 0.f;
        auto batch = [&] __device__(const int z0) {       // (first batch peeled: a loop header drains theta / m / v before its first request)
            float xs[DA_U];
#pragma unroll
            for (int u = 0; u < DA_U; ++u) xs[u] = slab[(size_t)min(z0 + u, ks - 1) * SP + col];
#pragma unroll
            for (int u = 0; u < DA_U; ++u)
                if (z0 + u < ks) t += xs[u];
        };
        if (ks > 0) batch(0);
        for (int z0 = DA_U; z0 < ks; z0 += DA_U) batch(z0);
        if (ltg_word_set(dead)) return;
        if (tail) {
            adam1(pe, me, ve, t, ad.lr_t, ad);
            p[e] = pe; m[e] = me; v[e] = ve;
        } else loss_out[0] = t;
        }
        if (!alone) return;
    }
    const int nb = alone ? 1 : gridDim.x - 1, b0 = alone ? 0 : blockIdx.x - 1;
    for (int e = b0 * NT + threadIdx.x; e < P4; e += nb * NT) {
        // (round 5: theta / m / v and the first eight slabs requested together, the slabs added in ascending order as before -- the plain
        // loop over a runtime slab count made every slab a round trip of its own: eight of them in a 5.7-us launch)
        ltg_f32x4 pp = ltg_ld4(p + 4 * e), mm = ltg_ld4(m + 4 * e), vv = ltg_ld4(v + 4 * e);
        ltg_f32x4 g = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < ks; z0 += DA_U) {
            ltg_f32x4 gs[DA_U];
#pragma unroll
            for (int u = 0; u < DA_U; ++u) gs[u] = *reinterpret_cast<const ltg_f32x4*>(slab + (size_t)min(z0 + u, ks - 1) * SP + 4 * e);
#pragma unroll
            for (int u = 0; u < DA_U; ++u)
                if (z0 + u < ks) g += gs[u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pp[j], mj = mm[j], vj = vv[j];
            adam1(pj, mj, vj, g[j], ad.lr_t, ad);
            pp[j] = pj; mm[j] = mj; vv[j] = vj;
        }
        if (ltg_word_set(dead)) return;
        *reinterpret_cast<ltg_f32x4*>(p + 4 * e) = pp;
        *reinterpret_cast<ltg_f32x4*>(m + 4 * e) = mm;
        *reinterpret_cast<ltg_f32x4*>(v + 4 * e) = vv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// small item slabs (I <= 4096): decoder layer 1, the row softmax + losses + dlogits, dh2, and the Adam tail
// ---------------------------------------------------------------------------------------------------------------------

// dec-1 (MultiVAE.py:169): logits = h2 . W_p1t^T + b_p1; operands rounded to bf16 when BF (LTG_PREC_BF16)
template <bool BF>
__global__ __launch_bounds__(NT) void fk_dec1(int B, int I, int H, const float* __restrict__ h2, const float* __restrict__ Wp1t,
                                              const float* __restrict__ bp1, float* __restrict__ logits) {
    __shared__ __attribute__((aligned(16))) float lds[Rg16::LDS_FLOATS];
    // the row tiles of one column tile (= the same 16 rows of W_p1t) meet in ONE XCD's L2: contiguous runs of the tile ids per XCD, column
    // tile major (round 5: dealt round-robin every row tile of a column fetched those rows into another L2 -- 21 MB of traffic for 3 MB)
    const LtgTile2 tl = xcd_tile2();
    const int m0 = tl.y * 16, n0 = tl.x * 16;
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(h2 + (size_t)m * H + k); };
    auto b_ld = [=] __device__(int, int k, int n) { return ltg_ld4(Wp1t + (size_t)n * H + k); };
    auto xf = [=] __device__(ltg_f32x4 v, int, int, int) { return BF ? ltg_bf16r4(v) : v; };
    const float biasv = bp1[min(n0 + (int)(threadIdx.x & 15), I - 1)];
    auto epi = [=] __device__(int, int m, int n, float v, bool ok) {
        if (ok) logits[(size_t)m * I + n] = v + biasv;
    };
    // (BF: both operands are bf16-rounded -- the product runs on the bf16 matrix pipe, 5 blocks of 32 per K slice instead of 10 of 16)
    if constexpr (BF) ltg_rgemm<1, 1, 1, 1, 4, 5, true>(B, I, H, m0, n0, a_ld, xf, b_ld, xf, epi, lds);
    else ltg_rgemm<1, 1, 1, 1, 4, 10>(B, I, H, m0, n0, a_ld, xf, b_ld, xf, epi, lds);
}

// One workgroup per user row: log-softmax statistics, the row's loss terms and dlogits in ONE pass (the row lives in
// registers).  train.py:145-157 + the closed form of SURVEY 8/a10:
//   dlogits[b][i] = p * (n_b / B + c * P_b) - x_bi / B - c * p * [(b, i) in S],  c = lambda / cnt * sum_j y_j
// rowout[b] = {neg_ll of the row, P_b = sum_{S_b} p, KL of the row, sum_j y_j}; the step's scalars are added up by the
// tail launch.  Needs no other row's statistics, so nothing has to meet between the forward and the backward.
// three sums and a maximum over the workgroup in one exchange (two barriers)
__device__ __forceinline__ void block_red4(float& a, float& b, float& c, float& mx, float (*red)[NT / 64]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = a;
        red[1][w] = b;
        red[2][w] = c;
        red[3][w] = mx;
    }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    mx = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
    __syncthreads();
}

__global__ __launch_bounds__(NT) void fk_row_dlogits(int B, int I, const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                     const float* __restrict__ values, const float* __restrict__ logits,
                                                     const float* __restrict__ kl_rows, const float* __restrict__ y, int nf,
                                                     const int32_t* __restrict__ cnt, float lam, const int32_t* __restrict__ f_row,
                                                     const int32_t* __restrict__ f_gen, const int32_t* __restrict__ f_pop,
                                                     float* __restrict__ dlog, float* __restrict__ lse, float* __restrict__ rowout) {
    __shared__ float s_l[RD_MAXI];      // the row's logits (the x . logit sum gathers from here)
    __shared__ float s_x[RD_MAXI];
    __shared__ uint8_t s_s[RD_MAXI];
    __shared__ float red[4][NT / 64];
    static_assert(NT / 64 == 4, "block_red4 adds four wave partials");
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (size_t)b * I;
    constexpr int PER = RD_MAXI / NT;
    const int e0 = indptr[b], e1 = indptr[b + 1];
    // every independent request first: the row, the fake tower's y, this thread's share of the fake-pair list
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + NT * j;
        v[j] = row[min(i, I - 1)];
    }
    // (round 5, second pass: ALSO up front -- the first batch of fake-pair triples, the thread's first sparse entry, cnt[0] and the row's KL
    // term: behind the barrier / at the end of the kernel each of them was a dependent round trip of its own -- seven in all in a 7.8-us launch)
    constexpr int RD_U = 4;
    int tg0[RD_U], tr0[RD_U], tp0[RD_U];
#pragma unroll
    for (int u = 0; u < RD_U; ++u) tg0[u] = tr0[u] = tp0[u] = -1;
    if (nf > 0) {      // (ONE uniform branch around the twelve requests: a select per element made a basic block -- and a wait -- of each)
#pragma unroll
        for (int u = 0; u < RD_U; ++u) {
            const int q = min(tid + u * NT, nf - 1);
            tg0[u] = f_gen[q];
            tr0[u] = f_row[q];
            tp0[u] = f_pop[q];
        }
    }
    const int cntv = cnt[0];
    const float klb = kl_rows[b];
    int it0 = -1;
    float x0 = 1.f;
    if (e0 + tid < e1) {
        it0 = indices[e0 + tid];
        if (values) x0 = values[e0 + tid];
    }
    // (round 5: the y's and the fake-pair triples of this thread in batches of RD_U requests, clamped and masked, consumed in the loop's
    // order -- as plain loops with a runtime bound every element was a round trip of its own: load, wait, use)
    float sy = 0.f;
    for (int q0 = tid; q0 < nf; q0 += RD_U * NT) {
        float ty[RD_U];
#pragma unroll
        for (int u = 0; u < RD_U; ++u) ty[u] = y[min(q0 + u * NT, nf - 1)];
#pragma unroll
        for (int u = 0; u < RD_U; ++u)
            if (q0 + u * NT < nf) sy += ty[u];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + NT * j;
        if (i < I) {
            s_l[i] = v[j];
            s_x[i] = 0.f;
            s_s[i] = 0;
            mx = fmaxf(mx, v[j]);
        } else v[j] = -INFINITY;
    }
    __syncthreads();
    float xl = 0.f, nx = 0.f;
    if (it0 >= 0) {      // (the entry requested up front, then the rest of a long row)
        s_x[it0] = x0;
        xl += x0 * s_l[it0];
        nx += x0;
    }
    for (int e = e0 + tid + NT; e < e1; e += NT) {
        const int it = indices[e];
        const float x = values ? values[e] : 1.f;
        s_x[it] = x;
        xl += x * s_l[it];
        nx += x;
    }
#pragma unroll
    for (int u = 0; u < RD_U; ++u)
        if (tid + u * NT < nf && tr0[u] == b && tg0[u] >= 0 && tg0[u] < I && tp0[u] >= 0) s_s[tg0[u]] = 1;
    for (int q0 = tid + RD_U * NT; q0 < nf; q0 += RD_U * NT) {
        int tg[RD_U], tr[RD_U], tp[RD_U];
#pragma unroll
        for (int u = 0; u < RD_U; ++u) {
            const int q = min(q0 + u * NT, nf - 1);
            tg[u] = f_gen[q];
            tr[u] = f_row[q];
            tp[u] = f_pop[q];
        }
#pragma unroll
        for (int u = 0; u < RD_U; ++u)
            if (q0 + u * NT < nf && tr[u] == b && tg[u] >= 0 && tg[u] < I && tp[u] >= 0) s_s[tg[u]] = 1;
    }
    block_red4(xl, nx, sy, mx, red);           // (its barrier also publishes s_x / s_s)
    float s = 0.f, psu = 0.f, zero = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + NT * j;
        const float ex = expf(v[j] - mx);      // exp(-inf) = 0 beyond I
        s += ex;
        psu += (i < I && s_s[i]) ? ex : 0.f;
    }
    block_red4(s, psu, zero, m2, red);
    const float l = mx + logf(s);
    const float ps = psu / s;                  // sum_{S_b} exp(logit - lse)
    const float invB = 1.f / (float)B, invs = 1.f / s;
    const float c = cntv > 0 ? lam / (float)cntv * sy : 0.f;
    const float alpha = nx * invB + c * ps;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + NT * j;
        if (i < I) {
            const float p = expf(v[j] - mx) * invs;
            dlog[(size_t)b * I + i] = p * alpha - s_x[i] * invB - (s_s[i] ? c * p : 0.f);
        }
    }
    if (tid == 0) {
        lse[b] = l;
        float* o = rowout + (size_t)b * 4;
        o[0] = -xl + nx * l;
        o[1] = ps;
        o[2] = klb;
        o[3] = sy;
    }
}

// da2 = (dlog . W_p1t) * (1 - h2^2)          [B][H], K = I
// (round 5: eight K slices, 512 threads; the 16 x 16 tile is finished by the first four waves -- 40 instead of 80 requests per wave; see fk_enc1)
constexpr int DH2_NT = 512;
template <bool BF>
__global__ __launch_bounds__(DH2_NT) void fk_dh2(int B, int I, int H, const float* __restrict__ dlog, const float* __restrict__ Wp1t,
                                             const float* __restrict__ h2, float* __restrict__ da2) {
    __shared__ __attribute__((aligned(16))) float lds[LtgRg<1, 1, 1, 1, 8>::LDS_FLOATS];
    const LtgTile2 tl = xcd_tile2();     // (as fk_dec1: a column block of W_p1t per XCD)
    const int m0 = tl.y * 16, n0 = tl.x * 16;
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld4(dlog + (size_t)m * I + k); };
    auto b_ld = [=] __device__(int, int k, int n) { return ltg_ld4s(Wp1t + (size_t)k * H + n, H); };
    auto xf = [=] __device__(ltg_f32x4 v, int, int, int) { return BF ? ltg_bf16r4(v) : v; };
    const float t = h2[(size_t)min(m0 + (int)(threadIdx.x >> 4), B - 1) * H + min(n0 + (int)(threadIdx.x & 15), H - 1)];
    auto epi = [=] __device__(int, int m, int n, float v, bool ok) {
        if (ok) da2[(size_t)m * H + n] = v * (1.f - t * t);
    };
    if constexpr (BF) ltg_rgemm<1, 1, 1, 1, 8, 4, true>(B, H, I, m0, n0, a_ld, xf, b_ld, xf, epi, lds);      // (4 blocks of 32 per slice: I <= 1 024 in one pass)
    else ltg_rgemm<1, 1, 1, 1, 8, 8>(B, H, I, m0, n0, a_ld, xf, b_ld, xf, epi, lds);
}

// "weight gradient + Adam" tile: G[m][n] = sum_k Lm(k, m) * Rm(k, n) over the K batch rows, fused with the TF-Adam update
// of W[m][n] (row stride ldw) -- theta / m / v of the tile are requested BEFORE the product.  ONES_L: an extra row m == Min
// of ones on the left (bias over n: MultiVAE.py b_q1, b_p0); otherwise an extra column n == Nin of ones on the right
// (bias over m: b_p1).  RND: operands rounded to bf16 (decoder layer 1 under LTG_PREC_BF16).  Nin % 4 == 0.
struct WgTensors {
    float *W, *mW, *vW, *b, *mb, *vb;
};
struct WgWhere {
    float *p, *m, *v;
    bool vec;
};
struct WgRegs {
    ltg_f32x4 p, m, v;
};
#ifndef LTG_TAIL_BN
#define LTG_TAIL_BN 32      // columns of a weight-gradient + Adam tile of fk_g_tail (32 or 64; rows: 32)
#endif
template <bool RND, bool ONES_L>
__device__ __forceinline__ void wgrad_adam_tile(int K, int Min, int Nin, const float* __restrict__ Lm, int ldl, const float* __restrict__ Rm,
                                                int ldr, WgTensors T, int ldw, AdamC ad, int m0, int n0, float* __restrict__ lds, unsigned dead = 0u) {
    const int M = ONES_L ? Min + 1 : Min, N = ONES_L ? Nin : Nin + 1;
    auto a_ld = [=] __device__(int, int m, int k) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Lm[(size_t)min(k + j, K - 1) * ldl + min(m, Min - 1)];
        return v;
    };
    auto a_xf = [=] __device__(ltg_f32x4 x, int, int m, int k) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = k + j < K ? ((ONES_L && m == Min) ? 1.f : (RND ? ltg_bf16r(x[j]) : x[j])) : 0.f;
        return v;
    };
    auto b_ld = [=] __device__(int, int k, int n) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Rm[(size_t)min(k + j, K - 1) * ldr + min(n, Nin - 1)];
        return v;
    };
    auto b_xf = [=] __device__(ltg_f32x4 x, int, int, int n) {
        ltg_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (!ONES_L && n == Nin) ? 1.f : (RND ? ltg_bf16r(x[j]) : x[j]);
        return v;
    };
    // where the float4 group (m, n .. n+3) lives: a weight row, or the bias vector (ONES_L: row Min), or -- for the ones
    // column n == Nin -- the single bias element of row m
    // (one base pointer + a per-lane offset: a per-lane select between two uniform pointers becomes a table in scratch)
    const ptrdiff_t dp = T.b - T.W, dm = T.mb - T.mW, dv = T.vb - T.vW;
    auto where = [=] __device__(int m, int n) {
        const bool wrow = ONES_L ? m < Min : n < Nin;
        const ptrdiff_t o = wrow ? (ptrdiff_t)m * ldw + n : (ONES_L ? (ptrdiff_t)n : (ptrdiff_t)m);
        WgWhere x;
        x.p = T.W + (wrow ? o : o + dp);
        x.m = T.mW + (wrow ? o : o + dm);
        x.v = T.vW + (wrow ? o : o + dv);
        x.vec = ONES_L || wrow;
        return x;
    };
    auto prefetch = [=] __device__(int m, int n, bool ok) {
        WgRegs r;
        r.p = r.m = r.v = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
        const WgWhere x = where(min(m, M - 1), ok ? n : 0);
        if (ok && x.vec) {
            r.p = ltg_ld4(x.p);
            r.m = ltg_ld4(x.m);
            r.v = ltg_ld4(x.v);
        } else if (ok) {
            r.p[0] = x.p[0];
            r.m[0] = x.m[0];
            r.v[0] = x.v[0];
        }
        return r;
    };
    auto epi4 = [=] __device__(WgRegs r, int m, int n, ltg_f32x4 g, bool ok) {
        if (!ok || ltg_word_set(dead)) return;      // (dead: the pipe's poison word, requested before anything else and first looked at here)
        const WgWhere x = where(m, n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = r.p[j], mj = r.m[j], vj = r.v[j];
            adam1(pj, mj, vj, g[j], ad.lr_t, ad);
            r.p[j] = pj; r.m[j] = mj; r.v[j] = vj;
        }
        if (x.vec) {
            *reinterpret_cast<ltg_f32x4*>(x.p) = r.p;
            *reinterpret_cast<ltg_f32x4*>(x.m) = r.m;
            *reinterpret_cast<ltg_f32x4*>(x.v) = r.v;
        } else {
            x.p[0] = r.p[0];
            x.m[0] = r.m[0];
            x.v[0] = r.v[0];
        }
    };
    // 32 x 32 tile, every wave the whole tile over a QUARTER of K (two 16-deep blocks at 100 batch rows): 32 operand registers per
    // lane instead of 56 (one 16 x 16 product over all of K per wave), so that six workgroups fit a CU and the ~1 600 tiles of an
    // Askubuntu-sized tail are resident in (almost) one round instead of two
    // (RND = the decoder's weight gradient under LTG_PREC_BF16: bf16-rounded operands on the bf16 matrix pipe, one 32-deep block per K slice)
    if constexpr (RND) ltg_rgemm_v4<2, LTG_TAIL_BN / 16, 1, 1, 4, 1, true>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, prefetch, epi4, lds);
    else ltg_rgemm_v4<2, LTG_TAIL_BN / 16, 1, 1, 4, 2>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, prefetch, epi4, lds);
}

// The Adam updates of the generator step as jobs riding with the backward chain (train.py:164; each is independent once
// every reader of its old weights has run).  One kernel, launched three times per step with different job sets:
//   launch A  dz tiles        + job 1 (W_p1t: dh2, the last reader of the old W_p1t, ran before)
//   launch B  dh1 tiles       + job 2 (W_p0: dz was its last reader)
//   launch C  jobs 3, 4, 5    (W_q1: dh1 was its last reader; W_q0 needs da1; the step's scalars)
//   job 1  dW_p1t + b_p1   (items x (H + 1), bf16-rounded operands under LTG_PREC_BF16)        -- small item slabs only
//   job 2  dW_p0 + b_p0    ((Z + 1) x H)          job 3  dW_q1 + b_q1   ((H + 1) x 2Z)
//   job 4  W_q0 + b_q0     dense float4 sweep, sparse gradient rows through slot[] (see k_enc0_bwd_adam)
//   job 5  the step's scalars from the per-row terms of fk_row_dlogits (train.py:154-157)         -- small item slabs only
struct TailArgs {
    int B, I, H, Z, nu;
    int nz, nh;                   // blocks of the dz / dh1 products riding in front (0 = not in this launch)
    int n1, n2, n3, n4, n5;       // blocks per job
    const float *Wp0, *Wq1, *mulv, *eps;
    float is_training;
    uint64_t seed, step;
    float *dmlv_out, *da1_out;
    const float *dlog, *h2, *z, *da2, *h1, *dmlv, *G;
    const float *xd, *da1;        // xd != NULL: job 4 = the dense product xd^T . da1 + Adam (no sparse rows, no slot map)
    const int32_t* slot;
    int q0_bias;                  // job 4 = only the bias row of the first encoder layer (lazy Adam clock: the item rows were updated by fk_enc0_grad)
    const float* rowout;
    const int32_t* cnt;
    float anneal, lam;
    float *loss_out, *loss_out2;
    // one-call step: `poison` != 0 -> nothing is updated; n_wait = 1: one more block at the end of the grid whose first thread polls for
    // `end_wait` (the clock slice on the side stream is done with every row: the next call's catch-up is the kernel behind this one)
    const unsigned* poison;
    int n_wait;
    LtgGate end_wait;
};
// (Round 5, measured: the launch holds 96 VGPRs + 16 AGPRs = four workgroups per CU, 1 024 slots for Askubuntu's 1 597 tiles.  Held to 5 / 6 / 7
// waves per SIMD with __launch_bounds__(NT, w) -- 92 VGPRs, 80 + 40 B of scratch, 72 + 100 B -- the G phase ran 77.2 / 78.5 / 83.0 against 76.6 ms
// per epoch: more resident tiles do not pay for fewer registers per tile.)
template <bool BF>
__global__ __launch_bounds__(NT) void fk_g_tail(TailArgs a, ltg_gen_state st, AdamC ad) {
    __shared__ __attribute__((aligned(16))) float lds[LtgRg<2, LTG_TAIL_BN / 16, 1, 1, 4>::LDS_FLOATS];
    int bid = blockIdx.x;
    if (a.n_wait && bid == (int)gridDim.x - 1) {
        if (threadIdx.x == 0) ltg_gate_wait_tail(a.end_wait);
        return;
    }
    // (round 5: the poison word is requested here and looked at in front of each job's first store -- as the guard of an early return it was a
    // round trip in front of every tile's requests)
    const unsigned dead = ltg_poison_word(a.poison);
    const int B = a.B, I = a.I, H = a.H, Z = a.Z;
    if (bid < a.n5) {   // job 5 FIRST in the grid (round 5): its chain -- row terms, three block sums, two more scalars, six stores -- started when the
                        // last tiles did and ended after them
        float* red = lds;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        const float sy = a.rowout[3];
        const int cntv = a.cnt[0];
        for (int b = threadIdx.x; b < B; b += NT) {
            x0 += a.rowout[(size_t)b * 4];
            x1 += a.rowout[(size_t)b * 4 + 1];
            x2 += a.rowout[(size_t)b * 4 + 2];
        }
        x0 = block_sum(x0, red);
        x1 = block_sum(x1, red);
        x2 = block_sum(x2, red);
        if (threadIdx.x == 0 && !ltg_word_set(dead)) {
            const float negll = x0 / (float)B, KL = x2 / (float)B;
            const float c = cntv > 0 ? a.lam / (float)cntv * sy : 0.f;
            const float vae = negll + a.anneal * KL, gan = -c * x1;
            const float r[6] = {vae + gan, vae, gan, x1, sy, c};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a.loss_out[i] = r[i];
                if (a.loss_out2) a.loss_out2[i] = r[i];
            }
        }
        return;
    }
    bid -= a.n5;
    // (The dz / dh1 tiles once rode in front of the jobs -- three launches of this kernel per step, measured +23 us; the variant
    // is gone: its 80 operand registers set the register count of the whole kernel and with it the tiles' occupancy.)
    // Every job's tiles are dealt to the XCDs in CONTIGUOUS runs (xcd_chunk on the job's own block index: blocks whose index agrees mod 8
    // share an XCD whatever the job's offset in the grid): an XCD then meets an eighth of the row tiles' operand columns (dlog, z, h1, xd) and
    // every column of the other operand, instead of all of both from all eight L2s (round 5; PMC traffic of the launch 67.3 MB against 39.25
    // algorithmic before).  Speed only: any placement computes the same values.
    if (bid < a.n1) {
        bid = xcd_chunk(bid, a.n1);
        const int tn = (H + 1 + LTG_TAIL_BN - 1) / LTG_TAIL_BN;
        const WgTensors T{st.p[3], st.m[3], st.v[3], st.p[7], st.m[7], st.v[7]};
        wgrad_adam_tile<BF, false>(B, I, H, a.dlog, I, a.h2, H, T, H, ad, (bid / tn) * 32, (bid % tn) * LTG_TAIL_BN, lds, dead);
        return;
    }
    bid -= a.n1;
    if (bid < a.n2) {
        bid = xcd_chunk(bid, a.n2);
        const int tn = (H + LTG_TAIL_BN - 1) / LTG_TAIL_BN;
        const WgTensors T{st.p[2], st.m[2], st.v[2], st.p[6], st.m[6], st.v[6]};
        wgrad_adam_tile<false, true>(B, Z, H, a.z, Z, a.da2, H, T, H, ad, (bid / tn) * 32, (bid % tn) * LTG_TAIL_BN, lds, dead);
        return;
    }
    bid -= a.n2;
    if (bid < a.n3) {
        bid = xcd_chunk(bid, a.n3);
        const int tn = (2 * Z + LTG_TAIL_BN - 1) / LTG_TAIL_BN;
        const WgTensors T{st.p[1], st.m[1], st.v[1], st.p[5], st.m[5], st.v[5]};
        wgrad_adam_tile<false, true>(B, H, 2 * Z, a.h1, H, a.dmlv, 2 * Z, T, 2 * Z, ad, (bid / tn) * 32, (bid % tn) * LTG_TAIL_BN, lds, dead);
        return;
    }
    bid -= a.n3;
    if (bid < a.n4 && a.xd) {
        bid = xcd_chunk(bid, a.n4);
        const int tn = (H + LTG_TAIL_BN - 1) / LTG_TAIL_BN;
        const WgTensors T{st.p[0], st.m[0], st.v[0], st.p[4], st.m[4], st.v[4]};
        wgrad_adam_tile<false, true>(B, I, H, a.xd, I, a.da1, H, T, H, ad, (bid / tn) * 32, (bid % tn) * LTG_TAIL_BN, lds, dead);
        return;
    }
    if (bid < a.n4 && a.q0_bias) {   // b_q0 from the partial bias rows of fk_enc0_grad + this step's learning rate into the clock's ring
        const int H4 = H >> 2;
        if (ltg_word_set(dead)) return;
        if (threadIdx.x == 0) st.q0_lr_hist[(st.q0_ord + 1) & (LTG_Q0_HIST - 1)] = ad.lr_t;
        float4* b4 = reinterpret_cast<float4*>(st.p[4]);
        float4* mb4 = reinterpret_cast<float4*>(st.m[4]);
        float4* vb4 = reinterpret_cast<float4*>(st.v[4]);
        const float4* G4 = reinterpret_cast<const float4*>(a.G);
        for (int c = threadIdx.x; c < H4; c += NT) {
            float4 g = G4[(size_t)a.nu * H4 + c];
#pragma unroll
            for (int j = 1; j < ENC0_BIAS_PARTS; ++j) {
                const float4 t = G4[(size_t)(a.nu + j) * H4 + c];
                g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
            }
            float4 p = b4[c], mm = mb4[c], vv = vb4[c];
            adam1(p.x, mm.x, vv.x, g.x, ad.lr_t, ad);
            adam1(p.y, mm.y, vv.y, g.y, ad.lr_t, ad);
            adam1(p.z, mm.z, vv.z, g.z, ad.lr_t, ad);
            adam1(p.w, mm.w, vv.w, g.w, ad.lr_t, ad);
            b4[c] = p;
            mb4[c] = mm;
            vb4[c] = vv;
        }
        return;
    }
    if (bid < a.n4) {
        const int H4 = H >> 2;
        if (ltg_word_set(dead)) return;
        const size_t total = (size_t)(I + 1) * H4;
        float4* W4 = reinterpret_cast<float4*>(st.p[0]);
        float4* m4 = reinterpret_cast<float4*>(st.m[0]);
        float4* v4 = reinterpret_cast<float4*>(st.v[0]);
        float4* b4 = reinterpret_cast<float4*>(st.p[4]);
        float4* mb4 = reinterpret_cast<float4*>(st.m[4]);
        float4* vb4 = reinterpret_cast<float4*>(st.v[4]);
        const float4* G4 = reinterpret_cast<const float4*>(a.G);
        for (size_t e = (size_t)bid * NT + threadIdx.x; e < total; e += (size_t)a.n4 * NT) {
            const int i = (int)(e / H4), c = (int)(e % H4);
            float4* P = i < I ? W4 + e : b4 + c;
            float4* Mm = i < I ? m4 + e : mb4 + c;
            float4* Vv = i < I ? v4 + e : vb4 + c;
            float4 p = *P, mm = *Mm, vv = *Vv;
            const int u = i < I ? a.slot[i] : a.nu;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u >= 0) g = G4[(size_t)u * H4 + c];
            if (i >= I) {
#pragma unroll
                for (int j = 1; j < ENC0_BIAS_PARTS; ++j) {
                    const float4 t = G4[(size_t)(a.nu + j) * H4 + c];
                    g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                }
            }
#define LTG_ADAM1(f) adam1(p.f, mm.f, vv.f, g.f, ad.lr_t, ad);
            LTG_ADAM1(x) LTG_ADAM1(y) LTG_ADAM1(z) LTG_ADAM1(w)
#undef LTG_ADAM1
            *P = p;
            *Mm = mm;
            *Vv = vv;
        }
        return;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wide discriminator, LTG_PREC_FP8 (BASELINE config 5): forward layers fed from OPERAND-FORMAT storage
// b_q0's Adam step with its gradient summed from da1 HERE, for the one-call step whose Adam tail runs on its own stream beside the
// sparse gradient kernel (fk_g_tail's bias job reads that kernel's partial bias rows; a kernel of its own so that its 16 rows in flight do
// not set fk_g_tail's register count).  Same bits as the partial rows: ENC0_BIAS_PARTS parts of `per` batch rows, each a serial sum from
// zero in the gradient kernel's light-row order (per <= G0_LIGHT, checked by the caller; a row past the end counts with weight 0), the
// parts added in ascending order.
// Shape: one thread per (float4 column, part) -- 32 columns x ENC0_BIAS_PARTS parts per workgroup, the part's rows requested at once, the
// parts added through LDS in ascending order (one thread walking all the parts took eight dependent round trips: 11.7 us on the tail
// stream, in front of the word the next call's enc-0 polls for).
constexpr int Q0B_COLS = NT / 8;
__global__ __launch_bounds__(NT) void fk_q0_bias_from_da1(int B, int H, const float* __restrict__ da1, ltg_gen_state st, AdamC ad,
                                                          const unsigned* __restrict__ poison) {
    static_assert(Q0B_COLS * 8 == NT && ENC0_BIAS_PARTS == 8, "one thread per (column, part)");
    __shared__ float4 parts[8][Q0B_COLS];
    if (ltg_poisoned(poison)) return;
    const int H4 = H >> 2, per = (B + ENC0_BIAS_PARTS - 1) / ENC0_BIAS_PARTS;
    const int cl = threadIdx.x % Q0B_COLS, pj = threadIdx.x / Q0B_COLS, c = blockIdx.x * Q0B_COLS + cl;
    const float4* D4 = reinterpret_cast<const float4*>(da1);
    {
        float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r0 = min(B, pj * per), r1 = min(B, (pj + 1) * per);
        float4 d[G0_LIGHT];
#pragma unroll
        for (int t = 0; t < G0_LIGHT; ++t) d[t] = D4[(size_t)min(r0 + t, B - 1) * H4 + min(c, H4 - 1)];
#pragma unroll
        for (int t = 0; t < G0_LIGHT; ++t) {
            const float sc = r0 + t < r1 ? 1.f : 0.f;
            sp.x = __builtin_fmaf(sc, d[t].x, sp.x); sp.y = __builtin_fmaf(sc, d[t].y, sp.y);
            sp.z = __builtin_fmaf(sc, d[t].z, sp.z); sp.w = __builtin_fmaf(sc, d[t].w, sp.w);
        }
        parts[pj][cl] = sp;
    }
    __syncthreads();
    if (pj != 0 || c >= H4) return;
    float4* b4 = reinterpret_cast<float4*>(st.p[4]);
    float4* mb4 = reinterpret_cast<float4*>(st.m[4]);
    float4* vb4 = reinterpret_cast<float4*>(st.v[4]);
    float4 p = b4[c], mm = mb4[c], vv = vb4[c];
    float4 g = parts[0][cl];
#pragma unroll
    for (int j = 1; j < ENC0_BIAS_PARTS; ++j) {
        const float4 sp = parts[j][cl];
        g.x += sp.x; g.y += sp.y; g.z += sp.z; g.w += sp.w;
    }
    adam1(p.x, mm.x, vv.x, g.x, ad.lr_t, ad);
    adam1(p.y, mm.y, vv.y, g.y, ad.lr_t, ad);
    adam1(p.z, mm.z, vv.z, g.z, ad.lr_t, ad);
    adam1(p.w, mm.w, vv.w, g.w, ad.lr_t, ad);
    b4[c] = p;
    mb4[c] = mm;
    vb4[c] = vv;
}

// ---------------------------------------------------------------------------------------------------------------------
// The fp8 mode of round 1 read every operand as fp32 and converted it on the way into LDS: 4 bytes moved per 1-byte operand,
// bound by L2 traffic (d_l1: 101 us for 11.5 GFLOP).  Here the operands LIVE in e4m3, k-contiguous: the frozen embedding table
// (emb_fp8 [F][h0], scale 2^8), transposed weight shadows (w1t [h1][h0], w2t [h2][h0], w3t [h3][h1+h2], scale 2^8; refreshed
// by the Adam sweep) and the branch layers' output (A1_fp8 [n][h1+h2], scale 2^6, written by the producing epilogue next to
// the fp32 copy the backward reads).  Same static scales and the same conversion (ltg_f2fp8) as before: the values the MFMA
// sees are bit-identical to the on-the-fly path, so the parity against the quantised oracle is unchanged.
//
// Block: 64 x 64 outputs per workgroup, each wave 32 x 32 over the whole K, no LDS: lane (r, q) requests 16 bytes
// k = 64 jb + 16 q .. + 15 of its row / column and the two MFMA steps of the block (v_mfma_f32_16x16x32_fp8_fp8) consume
// bytes 0-7 and 8-15 -- the k permutation both operands share.  Two register sets of four 64-byte blocks ping-pong: the
// requests of the next pass are in flight while this one multiplies.
// a_ld(t, m, k) / b_ld(t, k, n): PURE requests (t = which of the wave's two row / column tiles); a_mask(t) = all ones, or 0 to
// zero that row's operand (a hole in the pair list) -- applied when the registers are consumed, so that no arithmetic sits
// between the requests (a select next to its load makes the compiler wait for that load before issuing the next one).
template <class ALD, class AMK, class BLD, class EF>
__device__ __forceinline__ void ltg_rgemm8(int M, int N, int K, int m0, int n0, ALD a_ld, AMK a_mask, BLD b_ld, float scale, EF epi) {
    constexpr int NB = 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    int am[2], bn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        am[t] = min(m0 + (wm * 2 + t) * 16 + r, M - 1);
        bn[t] = min(n0 + (wn * 2 + t) * 16 + r, N - 1);
    }
    ltg_f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int nblk = K >> 6;   // K % 64 == 0
    const unsigned amk[2] = {a_mask(0), a_mask(1)};
    ltg_u32x4 a0[NB][2], b0[NB][2], a1[NB][2], b1[NB][2];
#define RG8_LOAD(A, B, base)                                                   \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                           \
        const int k = 64 * min((base) + i, nblk - 1) + 16 * q;                 \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                        \
            A[i][t] = a_ld(t, am[t], k);                                       \
            B[i][t] = b_ld(t, k, bn[t]);                                       \
        }                                                                      \
    }
#define RG8_MMA(A, B, base)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                                         \
        if ((base) + i < nblk) { /* wave-uniform */                                                                          \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                    \
                _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                             \
                    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) {                                                       \
                        const long av = (long)(((unsigned long)(A[i][tm][2 * s + 1] & amk[tm]) << 32) | (A[i][tm][2 * s] & amk[tm])); \
                        const long bv = (long)(((unsigned long)B[i][tn][2 * s + 1] << 32) | B[i][tn][2 * s]);               \
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av, bv, acc[tm][tn], 0, 0, 0);               \
                    }                                                                                                        \
        }                                                                                                                    \
    }
    RG8_LOAD(a0, b0, 0)
    for (int base = 0; base < nblk; base += 2 * NB) {
        RG8_LOAD(a1, b1, base + NB)
        __builtin_amdgcn_sched_barrier(0);
        RG8_MMA(a0, b0, base)
        __builtin_amdgcn_sched_barrier(0);
        RG8_LOAD(a0, b0, base + 2 * NB)
        __builtin_amdgcn_sched_barrier(0);
        RG8_MMA(a1, b1, base + NB)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef RG8_LOAD
#undef RG8_MMA
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int m = m0 + (wm * 2 + tm) * 16 + 4 * q + x, n = n0 + (wn * 2 + tn) * 16 + r;
                if (m < M && n < N) epi(tn, m, n, acc[tm][tn][x] * scale);
            }
}

__device__ __forceinline__ ltg_u32x4 ltg_ld16(const uint8_t* __restrict__ p) { return *reinterpret_cast<const ltg_u32x4*>(p); }
// tanh by one hardware exponential and one reciprocal (absolute error ~2e-7): the epilogues of the e4m3 layers, whose outputs are
// rounded to 3 mantissa bits anyway -- libm's tanhf is ~40 instructions per element and these epilogues are issue-bound (SQ
// counters: waves active 87 % of the time, MFMA busy 7 %)
__device__ __forceinline__ float ltg_tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

// branch layers from e4m3 storage: blockIdx.z = 0 popular -> h1, 1 niche -> h2
__global__ __launch_bounds__(NT) void fk8_d_l1(PairView pv, int h0, int h1, int h2, const uint8_t* __restrict__ emb8,
                                               const uint8_t* __restrict__ w1t8, const float* __restrict__ b1,
                                               const uint8_t* __restrict__ w2t8, const float* __restrict__ b2, DropView dA, DropView dB,
                                               float keep, uint64_t seed, uint64_t step, float* __restrict__ A1, uint8_t* __restrict__ A1_8) {
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    // 1-D grid of 8 x per x row-tiles blocks.  Blocks b, b + 8, ... share an XCD: XCD x takes the `per` consecutive COLUMN tiles
    // x per .. x per + per - 1 (of the tn1 + tn2 column tiles of both branches) for every row tile, so the weight rows an
    // XCD streams are 1/8 of the shadows (393 KB at the wide sizes) and stay in its 4-MiB L2 next to the embedding table --
    // with the natural order every XCD walked all 3 MB of weights + 2 MB of embeddings and was served from beyond its L2
    // (57 us; PMC).  Speed only.
    const int tn1 = (h1 + 63) / 64, tn2 = (h2 + 63) / 64, tm = (n + 63) / 64;
    const int per = (tn1 + tn2 + 7) / 8;
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int ct = x * per + i % per, rt = i / per;
    if (ct >= tn1 + tn2 || rt >= tm) return;
    const bool br = ct >= tn1;
    const int N = br ? h2 : h1;
    const int m0 = rt * 64, n0 = (br ? ct - tn1 : ct) * 64;
    const uint8_t* Wt = br ? w2t8 : w1t8;
    const float* bias = br ? b2 : b1;
    const int coff = br ? h1 : 0;
    // the embedding rows of this lane's two operand rows (ids requested once, not per k block)
    const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
    const uint8_t* erow[2];
    unsigned emask[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = min(m0 + (wm * 2 + t) * 16 + (lane & 15), n - 1);
        const int id = br ? pv.nic(m) : pv.pop(m);
        erow[t] = emb8 + (size_t)max(id, 0) * h0;
        emask[t] = id >= 0 ? 0xFFFFFFFFu : 0u;
    }
    auto a_ld = [=] __device__(int t, int, int k) { return ltg_ld16(erow[t] + k); };
    auto a_mask = [=] __device__(int t) { return emask[t]; };
    auto b_ld = [=] __device__(int, int k, int nn) { return ltg_ld16(Wt + (size_t)nn * h0 + k); };
    const int wn = (threadIdx.x >> 6) & 1;
    const float biasv[2] = {bias[min(n0 + (wn * 2) * 16 + (lane & 15), N - 1)], bias[min(n0 + (wn * 2 + 1) * 16 + (lane & 15), N - 1)]};   // requested up front
    auto epi = [=] __device__(int tn, int m, int nn, float v) {
        const float t = ltg_tanh_fast(v + biasv[tn]);
        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep) : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
        const float a = kp ? t / keep : 0.f;
        A1[(size_t)m * h12 + coff + nn] = a;
        A1_8[(size_t)m * h12 + coff + nn] = ltg_f2fp8(a * (float)(1 << FP8_S_ACT));
    };
    ltg_rgemm8(n, N, h0, m0, n0, a_ld, a_mask, b_ld, 1.f / (float)(1 << (FP8_S_EMB + FP8_S_W)), epi);
}

// fully connected layer from e4m3 storage
__global__ __launch_bounds__(NT) void fk8_d_l2(int n, int h12, int h3, const uint8_t* __restrict__ A1_8, const uint8_t* __restrict__ w3t8,
                                               const float* __restrict__ b3, DropView dC, float keep, uint64_t seed, uint64_t step,
                                               float* __restrict__ A3) {
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    auto a_ld = [=] __device__(int, int m, int k) { return ltg_ld16(A1_8 + (size_t)m * h12 + k); };
    auto a_mask = [=] __device__(int) { return 0xFFFFFFFFu; };
    auto b_ld = [=] __device__(int, int k, int nn) { return ltg_ld16(w3t8 + (size_t)nn * h12 + k); };
    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
    const float biasv[2] = {b3[min(n0 + (wn * 2) * 16 + (lane & 15), h3 - 1)], b3[min(n0 + (wn * 2 + 1) * 16 + (lane & 15), h3 - 1)]};
    auto epi = [=] __device__(int tn, int m, int nn, float v) {
        const float t = ltg_tanh_fast(v + biasv[tn]);
        A3[(size_t)m * h3 + nn] = dC.keep(m, nn, h3, seed, LTG_STREAM_D_DROP_C, step, keep) ? t / keep : 0.f;
    };
    ltg_rgemm8(n, h3, h12, m0, n0, a_ld, a_mask, b_ld, 1.f / (float)(1 << (FP8_S_ACT + FP8_S_W)), epi);
}

// ---- LDS-staged fp32 block for MANY pair rows (the batched fake towers of phase G: 10^5 rows per launch).  The 32 x 32
// register-resident tiles above are built for one round trip at ~2 000 rows; at 91 000 rows they re-fetch every operand per
// tile (2.9 GB from the L2s per tower).  64 x 64 outputs per workgroup, 32 floats of K per stage in LDS, global loads of the
// next stage in flight under the MFMAs of this one; each wave a 32 x 32 quarter with v_mfma_f32_16x16x4_f32.
//   a_row(r): start of operand row r of the tile (K floats, 16-byte aligned), nullptr = zero row; a_row(-1): any valid address.
//   B[k][n0 + c] = Bm[k * ldb + n0 + c] (row-major weights), columns >= N are zero.
template <class ARow, class EF>
__device__ __forceinline__ void ltg_sgemm32(int K, int N, int n0, ARow a_row, const float* __restrict__ Bm, int ldb, EF epi, float* __restrict__ lds) {
    constexpr int BM = 64, BN = 64, BK = 32, LDA = BK + 4, LDB = BN + 16;   // strides: conflict-free fragment reads (36 lr + lq, 16 lq + lr)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4, wm = w >> 1, wn = w & 1;
    float* As = lds;                         // [2][BM][LDA]
    float* Bs = lds + 2 * BM * LDA;          // [2][BK][LDB]
    const int arow = tid >> 3, akq = (tid & 7) * 4;          // A loader: rows arow + 32 j, floats akq .. akq + 3 of the stage
    const int bk = tid >> 6, bn = tid & 63;                  // B loader: k rows bk + 4 j, column bn
    const float* ap[2];
    bool aok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* q = a_row(arow + 32 * j);
        aok[j] = q != nullptr;
        ap[j] = (q ? q : a_row(-1)) + akq;
    }
    const bool bok = n0 + bn < N;
    const float* bp = Bm + min(n0 + bn, N - 1);
    ltg_f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[2];
    float rb[8];
#define SG32_FETCH(k0)                                                                                           \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const float4*>(ap[j] + min((k0), K - 4 - akq)); \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) rb[j] = bp[(size_t)min((k0) + bk + 4 * j, K - 1) * ldb]; \
    }
#define SG32_STASH(buf, k0)                                                                                      \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
            const bool ok = aok[j] && (k0) + akq < K;   /* K % 4 == 0: a float4 is inside or outside */          \
            *reinterpret_cast<float4*>(As + (size_t)((buf) * BM + arow + 32 * j) * LDA + akq) = ok ? ra[j] : make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                            \
            Bs[(size_t)((buf) * BK + bk + 4 * j) * LDB + bn] = (bok && (k0) + bk + 4 * j < K) ? rb[j] : 0.f;     \
    }
    SG32_FETCH(0)
    SG32_STASH(0, 0)
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;   // uniform
        if (more) SG32_FETCH(k0 + BK)
        const float* Aw = As + (size_t)(buf * BM + wm * 32 + lr) * LDA + lq;
        const float* Bw = Bs + (size_t)(buf * BK + lq) * LDB + wn * 32 + lr;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = Aw[(size_t)(i * 16) * LDA + kk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bw[(size_t)kk * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) SG32_STASH(buf ^ 1, k0 + BK)
        __syncthreads();
        buf ^= 1;
    }
#undef SG32_FETCH
#undef SG32_STASH
    epi(wm, wn, lr, lq, acc);
}
constexpr int SG32_LDS_FLOATS = 2 * 64 * 36 + 2 * 32 * 80;

// branch layers for many pair rows (forward only): blockIdx.x = 64-column tile over both branches, blockIdx.y = 64-row tile
__global__ __launch_bounds__(NT) void fks_d_l1(PairView pv, int h0, int h1, int h2, const float* __restrict__ emb, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, DropView dA,
                                               DropView dB, float keep, uint64_t seed, uint64_t step, float* __restrict__ A1) {
    __shared__ __attribute__((aligned(16))) float lds[SG32_LDS_FLOATS];
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const int tn1 = (h1 + 63) / 64;
    const bool br = (int)blockIdx.x >= tn1;
    const int N = br ? h2 : h1;
    const int m0 = blockIdx.y * 64, n0 = (br ? blockIdx.x - tn1 : blockIdx.x) * 64;
    const float* W = br ? w2 : w1;
    const float* bias = br ? b2 : b1;
    const int coff = br ? h1 : 0;
    auto a_row = [=] __device__(int r) -> const float* {
        if (r < 0 || m0 + r >= n) return r < 0 ? emb : nullptr;
        const int id = br ? pv.nic(m0 + r) : pv.pop(m0 + r);
        return id >= 0 ? emb + (size_t)id * h0 : nullptr;
    };
    auto epi = [=] __device__(int wm, int wn, int lr, int lq, ltg_f32x4 (&acc)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + wn * 32 + j * 16 + lr;
            const float bv = bias[min(nn, N - 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int m = m0 + wm * 32 + i * 16 + 4 * lq + x;
                    if (m < n && nn < N) {
                        const float t = tanhf(acc[i][j][x] + bv);
                        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep) : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
                        A1[(size_t)m * h12 + coff + nn] = kp ? t / keep : 0.f;
                    }
                }
        }
    };
    ltg_sgemm32(h0, N, n0, a_row, W, N, epi, lds);
}

// fully connected layer + the output unit's partial dot products for many pair rows (forward only: A3 is not kept):
// spart[(2 * tile + wave column)][row] = that 32-column strip's share of A3[row] . w4
__global__ __launch_bounds__(NT) void fks_d_l2(int n, int h12, int h3, const float* __restrict__ A1, const float* __restrict__ w3,
                                               const float* __restrict__ b3, const float* __restrict__ w4, DropView dC, float keep, uint64_t seed,
                                               uint64_t step, float* __restrict__ spart) {
    __shared__ __attribute__((aligned(16))) float lds[SG32_LDS_FLOATS];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    auto a_row = [=] __device__(int r) -> const float* { return (r < 0 || m0 + r >= n) ? (r < 0 ? A1 : nullptr) : A1 + (size_t)(m0 + r) * h12; };
    auto epi = [=] __device__(int wm, int wn, int lr, int lq, ltg_f32x4 (&acc)[2][2]) {
        float bv[2], wv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + wn * 32 + j * 16 + lr;
            bv[j] = b3[min(nn, h3 - 1)];
            wv[j] = nn < h3 ? w4[nn] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int m = m0 + wm * 32 + i * 16 + 4 * lq + x, mc = min(m, n - 1);
                float pd = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nn = n0 + wn * 32 + j * 16 + lr;
                    const float t = tanhf(acc[i][j][x] + bv[j]);
                    const bool kp = nn < h3 && dC.keep(mc, min(nn, h3 - 1), h3, seed, LTG_STREAM_D_DROP_C, step, keep);
                    pd += (kp ? t / keep : 0.f) * wv[j];
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) pd += __shfl_xor(pd, o);     // the 16 lanes lr of the strip
                if (lr == 0 && m < n) spart[(size_t)(2 * blockIdx.x + wn) * n + m] = pd;
            }
    };
    ltg_sgemm32(h12, h3, n0, a_row, w3, h3, epi, lds);
}

// ---- LDS-staged e4m3 block for the wide sizes.  The register-resident block above lets every wave fetch its own 32 operand
// rows: a 64 x 64 workgroup tile pulls each operand byte through the L1 twice and 696 such tiles move 356 MB from the L2s per
// branch-layer launch (7.4 TB/s at 48 us: L2-bandwidth-bound).  Here a workgroup owns BM x BN outputs, stages 128 bytes of K
// of both operands in LDS (16-byte global loads in flight under the MFMAs of the block before) and every wave multiplies its
// (BM / 2) x (BN / 2) quarter from there: 64 x 64 tiles move 178 MB for the same product (each operand byte once per workgroup),
// and three or four 37-KB workgroups per CU hide each other's load latency.  K % 128 == 0.
//   a_row(r) / b_row(c): start of operand row r / column c of the tile (k-contiguous e4m3), nullptr = all zero.
// The product loop, accumulators left in acc[BM / 32][BN / 32] (C layout of v_mfma_f32_16x16x32_fp8_fp8 per 16 x 16 block).
// Round 4: TWO K blocks of global loads in flight per workgroup (two register sets of 16-byte pieces; LDS stays double-buffered): a
// workgroup's stage used to last one L2 / HBM round trip (~1.2 us against 0.1 us of MFMA), 16 of them per 2048-deep tile.  The loop is
// unrolled by two with static set names, fetches are clamped instead of guarded and a block past the end is stashed as zeros (adds
// nothing), so the loop has no branch and every s_waitcnt is an exact count.  -DLTG_SG8_SHALLOW builds the one-block-ahead loop.
// (Round 5, measured and removed: THREE blocks in flight -- three register sets, the loop unrolled by six -- D step of config 5
// 128.1-128.5 against 115.2-115.5 us: more loads in flight make it slower, as larger tiles did; the block is not short of bytes in flight.)
constexpr int SG8_LDK = 128;   // bytes per LDS row of the staged e4m3 block (s8[2 * (BM + BN) * SG8_LDK] per workgroup)
template <int BM, int BN, class ARow, class BRow>
__device__ __forceinline__ void ltg_sgemm8_core(int K, ARow a_row, BRow b_row, ltg_f32x4 (&acc)[BM / 32][BN / 32], uint8_t* __restrict__ lds) {
    // LDS image (round 4): rows of 128 bytes WITHOUT padding, the sixteen 8-byte k-chunks of row r stored at chunk position c ^ (r & 15).
    // A fragment read is 16 lanes x 8 bytes of ONE logical chunk over 16 consecutive rows: with the 144-byte padded rows of before, rows r
    // and r + 8 met in the same banks (36 r mod 32 dwords repeats after 8 rows: SQ_LDS_BANK_CONFLICT 39 % of the LDS-active cycles, 23 % of
    // the wave cycles of fk8t_d_l1 waiting on LDS); swizzled, the 16 rows hit 16 different chunk positions = every bank once.  The loader's
    // 16-byte piece (two chunks of one row) stays one aligned 16-byte store, its halves exchanged in odd rows.
    constexpr int BK = 128, LDK = SG8_LDK, TM = BM / 32, TN = BN / 32, RA = BM / 32, RB = BN / 32;
    static_assert(LDK == BK, "unpadded rows: the swizzle replaces the pad");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4, wm = w >> 1, wn = w & 1;
    const int lrow = tid >> 3, lkc = (tid & 7) * 16;        // loader: rows lrow + 32 j, byte column lkc of the K block
    const int wcol = 16 * ((tid & 7) ^ ((lrow & 15) >> 1)); // ... stored at this byte column (rows lrow + 32 j share lrow & 15)
    const bool wodd = (lrow & 1) != 0;
    const uint8_t* ap[RA];
    const uint8_t* bp[RB];
    unsigned am[RA], bm[RB];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const uint8_t* q = a_row(lrow + 32 * j);
        am[j] = q ? 0xFFFFFFFFu : 0u;
        ap[j] = (q ? q : a_row(-1)) + lkc;     // a_row(-1): any valid address (masked to zero)
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const uint8_t* q = b_row(lrow + 32 * j);
        bm[j] = q ? 0xFFFFFFFFu : 0u;
        bp[j] = (q ? q : b_row(-1)) + lkc;
    }
    uint8_t* As = lds;                       // [2][BM][LDK]
    uint8_t* Bs = lds + 2 * BM * LDK;        // [2][BN][LDK]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    ltg_u32x4 ra[RA], rb[RB];
    // (macros, not lambdas: register arrays captured by reference end up in scratch)
#define SG8_FETCH(k0, XA, XB)                                                                              \
    {                                                                                                      \
        const int kf_ = min((k0), K - BK);                                                                 \
        _Pragma("unroll") for (int j = 0; j < RA; ++j) XA[j] = *reinterpret_cast<const ltg_u32x4*>(ap[j] + kf_); \
        _Pragma("unroll") for (int j = 0; j < RB; ++j) XB[j] = *reinterpret_cast<const ltg_u32x4*>(bp[j] + kf_); \
    }
#define SG8_STASH(buf, XA, XB, kk)                                                                         \
    {                                                                                                      \
        const unsigned in_ = (kk) < K ? 0xFFFFFFFFu : 0u;                                                  \
        _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                                   \
            const ltg_u32x4 x = XA[j];                                                                     \
            const unsigned mk = am[j] & in_;                                                               \
            ltg_u32x4 v;                                                                                   \
            v[0] = (wodd ? x[2] : x[0]) & mk; v[1] = (wodd ? x[3] : x[1]) & mk;                            \
            v[2] = (wodd ? x[0] : x[2]) & mk; v[3] = (wodd ? x[1] : x[3]) & mk;                            \
            *reinterpret_cast<ltg_u32x4*>(As + (size_t)((buf) * BM + lrow + 32 * j) * LDK + wcol) = v;     \
        }                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                                   \
            const ltg_u32x4 x = XB[j];                                                                     \
            const unsigned mk = bm[j] & in_;                                                               \
            ltg_u32x4 v;                                                                                   \
            v[0] = (wodd ? x[2] : x[0]) & mk; v[1] = (wodd ? x[3] : x[1]) & mk;                            \
            v[2] = (wodd ? x[0] : x[2]) & mk; v[3] = (wodd ? x[1] : x[3]) & mk;                            \
            *reinterpret_cast<ltg_u32x4*>(Bs + (size_t)((buf) * BN + lrow + 32 * j) * LDK + wcol) = v;     \
        }                                                                                                  \
    }
#define SG8_MFMA(buf)                                                                                      \
    {                                                                                                      \
        /* rows wm * (BM / 2) + 16 i + lr: r & 15 == lr; logical chunk ks / 8 + lq at chunk position (ks / 8 + lq) ^ lr */ \
        const uint8_t* Aw = As + (size_t)((buf) * BM + wm * (BM / 2) + lr) * LDK;                          \
        const uint8_t* Bw = Bs + (size_t)((buf) * BN + wn * (BN / 2) + lr) * LDK;                          \
        _Pragma("unroll") for (int ks = 0; ks < BK; ks += 32) {                                            \
            long af[TM], bf[TN];                                                                           \
            const int cx = 8 * (((ks >> 3) + lq) ^ lr);                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const long*>(Aw + (size_t)(i * 16) * LDK + cx); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const long*>(Bw + (size_t)(j * 16) * LDK + cx); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0); \
        }                                                                                                  \
    }
    SG8_FETCH(0, ra, rb)
    SG8_STASH(0, ra, rb, 0)
    __syncthreads();
#ifdef LTG_SG8_SHALLOW
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        SG8_FETCH(k0 + BK, ra, rb)       // the next block's loads fly under this block's MFMAs
        SG8_MFMA(buf)
        SG8_STASH(buf ^ 1, ra, rb, k0 + BK)     // the other buffer: its readers finished before the previous barrier
        __syncthreads();
        buf ^= 1;
    }
#else
    ltg_u32x4 ra2[RA], rb2[RB];
    SG8_FETCH(BK, ra, rb)                // block 1 in flight; block 2 follows inside the loop
    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
        SG8_FETCH(k0 + 2 * BK, ra2, rb2)
        SG8_MFMA(0)                              // block k0
        SG8_STASH(1, ra, rb, k0 + BK)            // (the other buffer: its readers finished before the previous barrier)
        __syncthreads();
        SG8_FETCH(k0 + 3 * BK, ra, rb)
        SG8_MFMA(1)                              // block k0 + BK (zeros past the end)
        SG8_STASH(0, ra2, rb2, k0 + 2 * BK)
        __syncthreads();
    }
#endif
#undef SG8_FETCH
#undef SG8_STASH
#undef SG8_MFMA
}

template <int BM, int BN, class ARow, class BRow, class EF>
__device__ __forceinline__ void ltg_sgemm8(int K, ARow a_row, BRow b_row, float scale, EF epi, uint8_t* __restrict__ lds) {
    constexpr int TM = BM / 32, TN = BN / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4, wm = w >> 1, wn = w & 1;
    ltg_f32x4 acc[TM][TN];
    ltg_sgemm8_core<BM, BN>(K, a_row, b_row, acc, lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int x = 0; x < 4; ++x) epi(wm * (BM / 2) + i * 16 + 4 * lq + x, wn * (BN / 2) + j * 16 + lr, acc[i][j][x] * scale);
}

// branch layers from e4m3 storage, LDS-staged: blockIdx.x = column tile over BOTH branches (popular -> h1, niche -> h2; XCD x keeps
// the column tiles x, x + 8, ...: its weight rows stay in its L2), blockIdx.y = row tile
template <int BM, int BN>
__global__ __launch_bounds__(NT) void fk8s_d_l1(PairView pv, int h0, int h1, int h2, const uint8_t* __restrict__ emb8,
                                                const uint8_t* __restrict__ w1t8, const float* __restrict__ b1,
                                                const uint8_t* __restrict__ w2t8, const float* __restrict__ b2, DropView dA, DropView dB,
                                                float keep, uint64_t seed, uint64_t step, float* __restrict__ A1, uint8_t* __restrict__ A1_8) {
    __shared__ __attribute__((aligned(16))) uint8_t s8[2 * (BM + BN) * SG8_LDK];
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const int tn1 = (h1 + BN - 1) / BN;
    const int ct = blockIdx.x, rt = blockIdx.y;
    const bool br = ct >= tn1;
    const int N = br ? h2 : h1;
    const int m0 = rt * BM, n0 = (br ? ct - tn1 : ct) * BN;
    const uint8_t* Wt = br ? w2t8 : w1t8;
    const float* bias = br ? b2 : b1;
    const int coff = br ? h1 : 0;
    auto a_row = [=] __device__(int r) -> const uint8_t* {
        if (r < 0 || m0 + r >= n) return r < 0 ? emb8 : nullptr;
        const int id = br ? pv.nic(m0 + r) : pv.pop(m0 + r);
        return id >= 0 ? emb8 + (size_t)id * h0 : nullptr;
    };
    auto b_row = [=] __device__(int c) -> const uint8_t* {
        if (c < 0) return Wt;
        return n0 + c < N ? Wt + (size_t)(n0 + c) * h0 : nullptr;
    };
    auto epi = [=] __device__(int r, int c, float v) {
        const int m = m0 + r, nn = n0 + c;
        if (m >= n || nn >= N) return;
        const float t = ltg_tanh_fast(v + bias[nn]);
        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep) : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
        const float a = kp ? t / keep : 0.f;
        A1[(size_t)m * h12 + coff + nn] = a;
        A1_8[(size_t)m * h12 + coff + nn] = ltg_f2fp8(a * (float)(1 << FP8_S_ACT));
    };
    ltg_sgemm8<BM, BN>(h0, a_row, b_row, 1.f / (float)(1 << (FP8_S_EMB + FP8_S_W)), epi, s8);
}

// fully connected layer from e4m3 storage, LDS-staged
template <int BM, int BN>
__global__ __launch_bounds__(NT) void fk8s_d_l2(int n, int h12, int h3, const uint8_t* __restrict__ A1_8, const uint8_t* __restrict__ w3t8,
                                                const float* __restrict__ b3, DropView dC, float keep, uint64_t seed, uint64_t step,
                                                float* __restrict__ A3) {
    __shared__ __attribute__((aligned(16))) uint8_t s8[2 * (BM + BN) * SG8_LDK];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    auto a_row = [=] __device__(int r) -> const uint8_t* { return (r < 0 || m0 + r >= n) ? (r < 0 ? A1_8 : nullptr) : A1_8 + (size_t)(m0 + r) * h12; };
    auto b_row = [=] __device__(int c) -> const uint8_t* { return (c < 0 || n0 + c >= h3) ? (c < 0 ? w3t8 : nullptr) : w3t8 + (size_t)(n0 + c) * h12; };
    auto epi = [=] __device__(int r, int c, float v) {
        const int m = m0 + r, nn = n0 + c;
        if (m >= n || nn >= h3) return;
        const float t = ltg_tanh_fast(v + b3[nn]);
        A3[(size_t)m * h3 + nn] = dC.keep(m, nn, h3, seed, LTG_STREAM_D_DROP_C, step, keep) ? t / keep : 0.f;
    };
    ltg_sgemm8<BM, BN>(h12, a_row, b_row, 1.f / (float)(1 << (FP8_S_ACT + FP8_S_W)), epi, s8);
}

// (re)build the e4m3 operand shadows of the discriminator from the fp32 tensors: emb8 [F][h0] and the TRANSPOSED weights
__global__ __launch_bounds__(NT) void k_d_shadow(int F, int h0, int h1, int h2, int h3, const float* __restrict__ emb, const float* __restrict__ w1,
                                                 const float* __restrict__ w2, const float* __restrict__ w3, uint8_t* __restrict__ emb8,
                                                 uint8_t* __restrict__ w1t8, uint8_t* __restrict__ w2t8, uint8_t* __restrict__ w3t8,
                                                 uint8_t* __restrict__ w3_8 = nullptr) {
    const size_t nE = (size_t)F * h0, n1 = (size_t)h0 * h1, n2 = (size_t)h0 * h2, n3 = (size_t)(h1 + h2) * h3;
    const size_t total = nE + n1 + n2 + n3;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        if (e < nE) emb8[e] = ltg_f2fp8(emb[e] * (float)(1 << FP8_S_EMB));
        else if (e < nE + n1) {
            const size_t i = e - nE, k = i / h1, nn = i % h1;
            w1t8[nn * h0 + k] = ltg_f2fp8(w1[i] * (float)(1 << FP8_S_W));
        } else if (e < nE + n1 + n2) {
            const size_t i = e - nE - n1, k = i / h2, nn = i % h2;
            w2t8[nn * h0 + k] = ltg_f2fp8(w2[i] * (float)(1 << FP8_S_W));
        } else {
            const size_t i = e - nE - n1 - n2, k = i / h3, nn = i % h3;
            w3t8[nn * (size_t)(h1 + h2) + k] = ltg_f2fp8(w3[i] * (float)(1 << FP8_S_W));
            if (w3_8) w3_8[i] = ltg_f2fp8(w3[i] * (float)(1 << FP8_S_W));
        }
    }
}
