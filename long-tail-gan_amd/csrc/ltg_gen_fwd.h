// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): generic generator forward kernels (MultiVAE.py:145-186): enc-0 as a sparse row gather, dense layers, row softmax.
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------
// Generator forward
// ---------------------------------------------------------------------------------------------

// enc-0 as a sparse row gather-sum (MultiVAE.py:148-155): h1 = tanh(dropout(l2norm(x)) . W_q0 + b).
// One 1024-thread workgroup per user row.  The row's (item, value*keep) list is staged through LDS;
// the 16 waves split the row's entries (so a 900-item history does not serialise on one wave), each
// lane owning float4 column chunks of the gathered W_q0 rows (coalesced 16-B loads); the wave
// partials meet in LDS.
constexpr int ENC_NT = 1024;
constexpr int ENC_NW = ENC_NT / 64;
__global__ __launch_bounds__(ENC_NT) void k_enc0_fwd(int H, int I, const int32_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                     const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed,
                                                     uint64_t step, const float* __restrict__ Wq0,
                                                     const float* __restrict__ bq0, float* __restrict__ h1,
                                                     float* __restrict__ row_scale, const float* __restrict__ row_norm2,
                                                     int item_lo, int Ig, int pre_only, int rps) {
    // item shard: `indices` are LOCAL item ids of this rank's slab [item_lo, item_lo + I); the dropout
    // RNG is keyed by the GLOBAL id so every shard draws the mask the unsharded run draws; row_norm2
    // (sum x^2 over the FULL row) replaces the local sum; pre_only writes the partial pre-activation
    // (no bias, no tanh) that the ranks all-reduce.
    extern __shared__ __attribute__((aligned(16))) float s_part[];  // [ENC_NW][H]
    __shared__ int s_idx[ENC_NT];
    __shared__ float s_val[ENC_NT];
    __shared__ float red[ENC_NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t kb = rps > 0 ? (uint64_t)(b % rps) : (uint64_t)b;   // several batches in one launch (ltg_fwd_opts.rows_per_step)
    step += rps > 0 ? (uint64_t)(b / rps) : 0;
    const int beg = indptr[b], end = indptr[b + 1];
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
    const float scale = 1.f / (keep * sqrtf(fmaxf(ss, 1e-12f)));  // l2_normalize eps, then /keep
    if (tid == 0) row_scale[b] = scale;
    const int H4 = H >> 2;
    constexpr int MAXQ = 4;  // H <= 1024
    float4 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
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
        }
        __syncthreads();
        const int cnt = min(ENC_NT, end - c0);
        // 4 entries per trip: their W_q0 row loads are independent, so 4 x MAXQ float4 loads are in flight
        for (int j = w; j < cnt; j += 4 * ENC_NW) {
            float v[4];
            const float4* wr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ju = j + u * ENC_NW;
                const bool ok = ju < cnt;
                v[u] = ok ? s_val[ju] : 0.f;
                wr[u] = reinterpret_cast<const float4*>(Wq0 + (size_t)s_idx[ok ? ju : j] * H);
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < H4) {
                    float4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = wr[u][c4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[q].x += v[u] * x[u].x;
                        acc[q].y += v[u] * x[u].y;
                        acc[q].z += v[u] * x[u].z;
                        acc[q].w += v[u] * x[u].w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < H4) reinterpret_cast<float4*>(s_part + (size_t)w * H)[c4] = acc[q];
    }
    __syncthreads();
    for (int c = tid; c < H; c += ENC_NT) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < ENC_NW; ++i) t += s_part[(size_t)i * H + c];
        h1[(size_t)b * H + c] = pre_only ? t * scale : tanhf(t * scale + bq0[c]);
    }
}

// h1 = tanh(h1_pre + b) after the partial pre-activations of the item shards were all-reduced
__global__ __launch_bounds__(NT) void k_bias_tanh(int n, int H, const float* __restrict__ bias, float* __restrict__ h) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) h[i] = tanhf(h[i] + bias[i % H]);
}

// float4 of 4 consecutive floats at p[i .. i+3], zero where i + j >= n (n % 4 == 0 at every call site, so a group is
// either whole or absent; the address is clamped, the load unconditional)
__device__ __forceinline__ float4 ltg_ld4(const float* __restrict__ p, int i, int n, bool ok) {
    const float4 v = *reinterpret_cast<const float4*>(p + min(i, n - 4));
    const bool k = ok && i < n;
    return make_float4(k ? v.x : 0.f, k ? v.y : 0.f, k ? v.z : 0.f, k ? v.w : 0.f);
}

// Generic dense layer  C = act(A[M][K] . B[K][N] + bias)  (fp32 MFMA); act: 0 none, 1 tanh.
// Serves enc-1 (MultiVAE.py:152) and dec-0 (MultiVAE.py:168-172).
template <int ACT, bool V, int BKV = 128>
__global__ __launch_bounds__(NT) void k_dense_fwd(int M, int N, int K, const float* __restrict__ A,
                                                  const float* __restrict__ Bw, const float* __restrict__ bias,
                                                  float* __restrict__ C) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    auto a = [=] __device__(int m, int k) -> float { return A[(size_t)m * K + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Bw[(size_t)k * N + n]; };
    auto epi = [=] __device__(int m, int n, float acc) {
        const float x = acc + bias[n];
        C[(size_t)m * N + n] = ACT == 1 ? tanhf(x) : x;
    };
    if constexpr (V) {   // 16-B loads (K % 4 == 0, N % 4 == 0): same MFMA sequence, a quarter of the load instructions
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(A + (size_t)min(m, M - 1) * K, k, K, m < M); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Bw + (size_t)min(k, K - 1) * N, n, N, k < K); };
        ltg_gemm_block<false, 32, 32, BKV, 2, 2, false, true, false, 0, 0, 3>(M, N, m0, n0, 0, K, a4, b4, epi);
    } else {
        ltg_gemm_block<false, 32, 32, 128, 2, 2, false, true>(M, N, m0, n0, 0, K, a, b, epi);
    }
}

// Reparameterisation + KL (MultiVAE.py:157-162, :178-181).
__global__ __launch_bounds__(NT) void k_reparam(int Z, const float* __restrict__ mulv, const float* __restrict__ eps_in,
                                                float is_training, uint64_t seed, uint64_t step,
                                                float* __restrict__ z, float* __restrict__ kl_rows) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    float kl = 0.f;
    for (int j = threadIdx.x; j < Z; j += NT) {
        const float mu = mulv[(size_t)b * 2 * Z + j], lv = mulv[(size_t)b * 2 * Z + Z + j];
        const float sd = expf(0.5f * lv);
        kl += 0.5f * (-lv + expf(lv) + mu * mu - 1.f);
        float e = 0.f;
        if (is_training != 0.f)
            e = eps_in ? eps_in[(size_t)b * Z + j] : ltg_rng_normal(seed, LTG_STREAM_VAE_EPS, step, (uint64_t)b * Z + j);
        z[(size_t)b * Z + j] = mu + is_training * e * sd;
    }
    kl = block_sum(kl, red);
    if (threadIdx.x == 0) kl_rows[b] = kl;
}

// dec-1 (MultiVAE.py:169): logits[b][i] = h2[b][:] . W_p1t[i][:] + b_p1[i]; the big GEMM.
template <bool BF16, bool BIG, bool V = false>
__global__ __launch_bounds__(NT) void k_dec1_fwd(int M, int I, int H, const float* __restrict__ h2,
                                                 const float* __restrict__ Wp1t, const float* __restrict__ bp1,
                                                 float* __restrict__ logits) {
    // BIG: 128 x 64 tiles (a whole training batch per tile: W_p1t leaves HBM once); small item counts
    // use 32 x 32 tiles to spread the few tiles over more CUs.
    constexpr int BM = BIG ? 128 : 32, BN = BIG ? 64 : 32;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    auto a = [=] __device__(int m, int k) -> float { return h2[(size_t)m * H + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Wp1t[(size_t)n * H + k]; };
    auto epi = [=] __device__(int m, int n, float acc) { logits[(size_t)m * I + n] = acc + bp1[n]; };
    if constexpr (V) {   // 16-B loaders (H % 4 == 0)
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(h2 + (size_t)min(m, M - 1) * H, k, H, m < M); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Wp1t + (size_t)min(n, I - 1) * H, k, H, n < I); };
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, false, false, 0, 0, 3>(M, I, m0, n0, 0, H, a4, b4, epi);
    } else {
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, false>(M, I, m0, n0, 0, H, a, b, epi);
    }
}
