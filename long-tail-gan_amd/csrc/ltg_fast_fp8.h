// Part of csrc/ltg_fast.h (included there, in this order, inside ltg_kernels.hip's anonymous namespace): the wide discriminator's fp8 forward kernels fed from operand-format storage, the LDS-staged fp32 forward-only tower (fks_*) and the e4m3 block ltg_sgemm8_core.
// Split out of the 2 100-line header in round 6 -- the code is unchanged.
#pragma once

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
