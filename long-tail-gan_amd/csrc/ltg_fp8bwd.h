// Backward of the wide discriminator (BASELINE config 5: h0-h3 = 2048/1024/512/256, discriminator.py:23-55) with e4m3 GEMM operands
// in OPERAND FORMAT: every backward product reads k-contiguous e4m3 bytes through the LDS-staged block ltg_sgemm8 instead of
// converting fp32 values (and gathering embedding rows element by element) inside the GEMM loaders.
//
// The three backward GEMMs contract over different dimensions, so each activation / gradient is written once in every
// orientation a consumer needs, by the kernel that produces it (pair rows padded with zeros to NP = a multiple of 128):
//
//   producer                      writes (e4m3, the mode's static scales)                       consumed by
//   fk8t_d_l1                     A1_8   [n][h12]      (k = h12 contiguous)                     fc layer forward (as before)
//                                 A1T_8  [h12 + 1][NP] (k = pair rows; row h12 = ones)          dw3 / db3 = A1^T . dpre3
//                                 dA1T_16 [h12][NP]    (bf16: d A1 / d pre, transposed)         job A's epilogue: dpre1 = dA1 * it (no fp32 A1)
//   k8_d_out                      dpre3_8  [n][h3]     (k = h3)                                 dpre1 = dpre3 . w3^T
//                                 dpre3T_8 [h3][NP]    (k = pair rows)                          dw3 / db3
//   k8_d_bwd1 job A epilogue      dpre1T_8 [h12][NP]   (k = pair rows)                          dw1 / dw2 = E^T . dpre1
//   k8_gather_t                   ET_8 [2][h0 + 1][NP] (gathered embedding rows transposed;     dw1 / db1, dw2 / db2
//                                                       row h0 = ones)
//   ltg_refresh_d_shadow / Adam   w3_8 [h12][h3]       (k = h3: w3 in its own layout)           dpre1 = dpre3 . w3^T
//
// Same static scales and the same conversion (ltg_f2fp8 of x * 2^S) as the on-the-fly path, so the MFMAs see identical operand
// values; only the order of the fp32 additions differs.  The split-K chunks are D8_KCHUNK pair rows (1024) instead of 256: the fp8
// products are short, and the Adam sweep then reads two or three gradient slabs instead of eight.
// Included by ltg_kernels.hip inside its anonymous namespace, after ltg_fast.h.
#pragma once

constexpr int D8_KCHUNK = 1024;
inline int d8_np(int n) { return (n + 127) / 128 * 128; }

struct LtgNoQuad {
    __device__ __forceinline__ void operator()(int, int, const float*) const {}
};

// ltg_sgemm8 with a second epilogue per accumulator register group: epiq(r0, c, v[4]) = the four values of rows r0 .. r0 + 3 of
// column c (the MFMA's C layout) -- what a TRANSPOSED byte store wants (four consecutive bytes of one output row).
template <int BM, int BN, class ARow, class BRow, class EF, class EQ>
__device__ __forceinline__ void ltg_sgemm8q(int K, ARow a_row, BRow b_row, float scale, EF epi, EQ epiq, uint8_t* __restrict__ lds) {
    constexpr int TM = BM / 32, TN = BN / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4, wm = w >> 1, wn = w & 1;
    ltg_f32x4 acc[TM][TN];
    ltg_sgemm8_core<BM, BN>(K, a_row, b_row, acc, lds);      // (csrc/ltg_fast.h: the product loop both blocks share)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                v[x] = acc[i][j][x] * scale;
                epi(wm * (BM / 2) + i * 16 + 4 * lq + x, wn * (BN / 2) + j * 16 + lr, v[x]);
            }
            epiq(wm * (BM / 2) + i * 16 + 4 * lq, wn * (BN / 2) + j * 16 + lr, v);
        }
}

__device__ __forceinline__ unsigned ltg_pack4_fp8(const float* v, float scale) {
    return (unsigned)ltg_f2fp8(v[0] * scale) | ((unsigned)ltg_f2fp8(v[1] * scale) << 8) | ((unsigned)ltg_f2fp8(v[2] * scale) << 16) |
           ((unsigned)ltg_f2fp8(v[3] * scale) << 24);
}

// branch layers from e4m3 storage (fk8s_d_l1) that ALSO leave A1 transposed in e4m3 for the backward: row tiles cover the padded
// row count NP (rows >= n are written as zeros), the column-tile-0 workgroups add the ones row of the bias gradient.
template <int BM, int BN>
__global__ __launch_bounds__(NT) void fk8t_d_l1(PairView pv, int h0, int h1, int h2, int NP, const uint8_t* __restrict__ emb8,
                                                const uint8_t* __restrict__ w1t8, const float* __restrict__ b1,
                                                const uint8_t* __restrict__ w2t8, const float* __restrict__ b2, DropView dA, DropView dB,
                                                float keep, uint64_t seed, uint64_t step, unsigned short* __restrict__ dA1T, uint8_t* __restrict__ A1_8,
                                                uint8_t* __restrict__ A1T_8) {
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
    if (ct == 0 && threadIdx.x < BM) {   // ones row (bias gradient db3 rides in the dw3 product): 1 * 2^S_ACT for real rows
        const int m = m0 + threadIdx.x;
        if (m < NP) A1T_8[(size_t)h12 * NP + m] = m < n ? ltg_f2fp8((float)(1 << FP8_S_ACT)) : (uint8_t)0;
    }
    auto a_row = [=] __device__(int r) -> const uint8_t* {
        if (r < 0 || m0 + r >= n) return r < 0 ? emb8 : nullptr;
        const int id = br ? pv.nic(m0 + r) : pv.pop(m0 + r);
        return id >= 0 ? emb8 + (size_t)id * h0 : nullptr;
    };
    auto b_row = [=] __device__(int c) -> const uint8_t* {
        if (c < 0) return Wt;
        return n0 + c < N ? Wt + (size_t)(n0 + c) * h0 : nullptr;
    };
    auto act = [=] __device__(int m, int nn, float v) -> float {
        const float t = ltg_tanh_fast(v + bias[nn]);
        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep) : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
        return kp ? t / keep : 0.f;
    };
    // The tile leaves through LDS (round 4): every lane used to issue 36 stores -- 16 fp32 values in 64-byte runs, 16 single bytes, four 4-byte
    // pieces of the transposed copy each to its own cache line -- and a build without them ran 21.9 instead of 29.8 us.  Now the activations
    // go into a [BM][BN + 4] fp32 image in the staging buffer (free behind the product's last barrier) and leave as whole 16-byte pieces:
    // 256-byte rows of A1, 64-byte rows of A1_8, 64-byte columns of A1T_8 (rows >= n as zeros).  Same values, same conversions.
    static_assert(BM == 64 && BN == 64 && NT == 256, "one 16-byte piece per thread and copy");
    constexpr int LT = BN + 4;
    static_assert(BM * LT * 4 <= 2 * (BM + BN) * SG8_LDK, "the fp32 image fits the staging buffer");
    float* T = reinterpret_cast<float*>(s8);
    {
        constexpr int TM = BM / 32, TN = BN / 32;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4, wm = w >> 1, wn = w & 1;
        ltg_f32x4 acc[TM][TN];
        ltg_sgemm8_core<BM, BN>(h0, a_row, b_row, acc, s8);      // (ends behind a barrier: nobody reads the staged operands any more)
        const float scale = 1.f / (float)(1 << (FP8_S_EMB + FP8_S_W));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = wn * (BN / 2) + j * 16 + lr, nn = n0 + c;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int r = wm * (BM / 2) + i * 16 + 4 * lq + x, m = m0 + r;
                    T[r * LT + c] = (m < n && nn < N) ? act(m, nn, acc[i][j][x] * scale) : 0.f;
                }
            }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const float s_act = (float)(1 << FP8_S_ACT);
    if (n0 + BN <= N) {   // (every layer size is a multiple of 64 in this mode: whole column tiles)
        {                                                              // A1_8: 16-byte pieces, 4 per row
            const int r = tid >> 2, p = tid & 3;
            if (m0 + r < n) {
                ltg_u32x4 o;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = T[r * LT + 16 * p + 4 * g + x];
                    o[g] = ltg_pack4_fp8(v, s_act);
                }
                *reinterpret_cast<ltg_u32x4*>(A1_8 + (size_t)(m0 + r) * h12 + coff + n0 + 16 * p) = o;
            }
        }
        {                                                              // A1T_8: column c, rows 16 p .. 16 p + 15 (a wave = one p: lanes walk the columns)
            const int c = tid & 63, p = tid >> 6;
            if (m0 + 16 * p < NP) {
                ltg_u32x4 o, d0, d1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = T[(16 * p + 4 * g + x) * LT + c];
                    o[g] = ltg_pack4_fp8(v, s_act);
                    // ... and the factor the backward multiplies dA1 with, d A1 / d pre = dropout' tanh' (0 for dropped elements and the padded
                    // rows), as bf16 in the SAME transposed layout: 32 bytes per thread instead of the 64 of an fp32 copy of A1 (round 5)
                    const unsigned q0 = (unsigned)ltg_f2bf(dact(v[0], keep)) | ((unsigned)ltg_f2bf(dact(v[1], keep)) << 16);
                    const unsigned q1 = (unsigned)ltg_f2bf(dact(v[2], keep)) | ((unsigned)ltg_f2bf(dact(v[3], keep)) << 16);
                    if (g < 2) { d0[2 * g] = q0; d0[2 * g + 1] = q1; } else { d1[2 * (g - 2)] = q0; d1[2 * (g - 2) + 1] = q1; }
                }
                *reinterpret_cast<ltg_u32x4*>(A1T_8 + (size_t)(coff + n0 + c) * NP + m0 + 16 * p) = o;
                ltg_u32x4* dd = reinterpret_cast<ltg_u32x4*>(dA1T + (size_t)(coff + n0 + c) * NP + m0 + 16 * p);
                dd[0] = d0;
                dd[1] = d1;
            }
        }
    } else {              // a ragged column tile: element by element
        for (int e = tid; e < BM * BN; e += NT) {
            const int r = e / BN, c = e % BN, m = m0 + r, nn = n0 + c;
            if (nn >= N || m >= NP) continue;
            const float a = T[r * LT + c];
            if (m < n) A1_8[(size_t)m * h12 + coff + nn] = ltg_f2fp8(a * s_act);
            A1T_8[(size_t)(coff + nn) * NP + m] = ltg_f2fp8(a * s_act);
            dA1T[(size_t)(coff + nn) * NP + m] = ltg_f2bf(dact(a, keep));
        }
    }
}

// Output unit + loss terms + the gradient into the fc layer's pre-activation (k_d_out<true>, discriminator.py:45,55, train.py:142)
// for a tile of 16 pair rows, dpre3 = ds * w4 * dact(A3) written ONLY in e4m3: row-major [n][h3] and transposed [h3][NP] (through an
// LDS tile: 16-byte runs).  Rows in [n, NP) come out as zeros.  h3 <= 512.  grid NP / 16.
constexpr int D8_OUT_LD = 80, D8_OUT_CM = 8;   // h3 <= 64 * D8_OUT_CM
__global__ __launch_bounds__(NT) void k8_d_out(PairView pv, int h3, int NP, const float* __restrict__ A3, const float* __restrict__ w4,
                                               const float* __restrict__ b4, float keep, float* __restrict__ y, float* __restrict__ ds,
                                               float* __restrict__ lrow, uint8_t* __restrict__ dpre3_8, uint8_t* __restrict__ dpre3T_8) {
    __shared__ __attribute__((aligned(16))) uint8_t T[64 * D8_OUT_CM * D8_OUT_LD];
    const int n = pv.nr + pv.nf, m0 = blockIdx.x * 16;          // 16 pair rows per workgroup: wave w owns rows 4 w .. 4 w + 3
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ik = 1.f / keep, b4v = b4[0];
    // every operand of the wave's four rows is requested before anything is consumed (one round trip)
    float a[4][D8_OUT_CM], wv[D8_OUT_CM];
#pragma unroll
    for (int j = 0; j < D8_OUT_CM; ++j) {
        const int c = min(lane + 64 * j, h3 - 1);
        wv[j] = lane + 64 * j < h3 ? w4[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][j] = A3[(size_t)min(m0 + 4 * w + i, n - 1) * h3 + c];
    }
    int pid[4], nid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(m0 + 4 * w + i, n - 1);
        pid[i] = pv.pop(r);
        nid[i] = pv.nic(r);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 4 * w + i, r = m0 + rl;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < D8_OUT_CM; ++j) s += a[i][j] * wv[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        s += b4v;
        const float yy = 1.f / (1.f + expf(-s));
        const bool ok = r < n && pid[i] >= 0 && nid[i] >= 0, real = r < pv.nr;
        const float dsr = ok ? (real ? -(1.f - yy) : yy) : 0.f;
        if (lane == 0 && r < n) {
            y[r] = ok ? yy : 0.f;
            ds[r] = dsr;
            lrow[r] = ok ? (real ? -logf(yy) : -logf(1.f - yy)) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < D8_OUT_CM; ++j) {
            const int c = lane + 64 * j;
            if (c < h3) {
                const float av = a[i][j], t = av * keep;
                const float v = (r < n && av != 0.f) ? dsr * wv[j] * (1.f - t * t) * ik : 0.f;   // (k_d_out's expression)
                const uint8_t q = ltg_f2fp8(v * (float)(1 << FP8_S_G3));
                if (r < n) dpre3_8[(size_t)r * h3 + c] = q;
                T[c * D8_OUT_LD + rl] = q;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < h3; c += NT)   // 16 rows of column c: one 16-byte run of the transposed copy
        *reinterpret_cast<ltg_u32x4*>(dpre3T_8 + (size_t)c * NP + m0) = *reinterpret_cast<const ltg_u32x4*>(T + c * D8_OUT_LD);
}

// The embedding rows of the pair batch, gathered and TRANSPOSED in e4m3: ET[br][c][m] = emb8[id_br(m)][c] (0 for holes and rows
// >= n), row h0 = ones (1 * 2^S_EMB for rows < n: the bias gradients db1 / db2 ride in the dw1 / dw2 products, also for pairs with
// a hole, like the on-the-fly path).  Depends on the pair ids only.  grid (h0 / 64, NP / 64, 2).
__global__ __launch_bounds__(NT) void k8_gather_t(PairView pv, int h0, int NP, const uint8_t* __restrict__ emb8, uint8_t* __restrict__ ET) {
    __shared__ __attribute__((aligned(16))) uint8_t T[64 * 80];
    const int n = pv.nr + pv.nf;
    const int c0 = blockIdx.x * 64, m0 = blockIdx.y * 64, br = blockIdx.z;
    uint8_t* out = ET + (size_t)br * (h0 + 1) * NP;
    const int rl = threadIdx.x >> 2, piece = threadIdx.x & 3;
    const int r = m0 + rl;
    int id = -1;
    if (r < n) id = br ? pv.nic(r) : pv.pop(r);
    ltg_u32x4 v = ltg_u32x4{0u, 0u, 0u, 0u};
    if (id >= 0) v = *reinterpret_cast<const ltg_u32x4*>(emb8 + (size_t)id * h0 + c0 + 16 * piece);
    *reinterpret_cast<ltg_u32x4*>(T + rl * 80 + 16 * piece) = v;
    __syncthreads();
    {   // thread -> (column cl, 16 rows of it)
        const int cl = threadIdx.x >> 2, part = threadIdx.x & 3;
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) x |= (unsigned)T[(16 * part + 4 * j + b) * 80 + cl] << (8 * b);
            o[j] = x;
        }
        *reinterpret_cast<ltg_u32x4*>(out + (size_t)(c0 + cl) * NP + m0 + 16 * part) = ltg_u32x4{o[0], o[1], o[2], o[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) out[(size_t)h0 * NP + m0 + threadIdx.x] = m0 + (int)threadIdx.x < n ? ltg_f2fp8((float)(1 << FP8_S_EMB)) : (uint8_t)0;
}

// Backward stage 1 from operand-format storage, ONE launch, three jobs by block index (k_d_bwd1's jobs; SP = stride of a gradient
// slab, a multiple of 4 floats: the Adam sweep reads the slabs in 16-byte pieces):
//   job A  dpre1 = (dpre3 . w3^T) * dact(A1)  -> dpre1T_8 [h12][NP] (e4m3, transposed; rows >= n zero)      64 x 64 tiles over (NP, h12)
//   job B  slab[z] = A1^T . dpre3 (+ ones row -> db3) over the pair rows of chunk z                          64 x 64 tiles over (h12 + 1, h3)
//   job C  slab[z]: dw4 = A3^T . ds, db4 = sum ds                                                            fp32 column sums
__global__ __launch_bounds__(NT) void k8_d_bwd1(int n, int NP, int h12, int h3, int nA, int nB, DLayout L, int SP, const unsigned short* __restrict__ dA1T,
                                                const float* __restrict__ A3, const float* __restrict__ ds,
                                                const uint8_t* __restrict__ dpre3_8, const uint8_t* __restrict__ dpre3T_8,
                                                const uint8_t* __restrict__ A1T_8, const uint8_t* __restrict__ w3_8, float keep,
                                                uint8_t* __restrict__ dpre1T_8, float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) uint8_t s8[2 * (64 + 64) * SG8_LDK];
    int bid = blockIdx.x;
    if (bid < nA) {
        const int tn = (h12 + 63) / 64;
        const int m0 = (bid / tn) * 64, n0 = (bid % tn) * 64;
        auto a_row = [=] __device__(int r) -> const uint8_t* { return (r < 0 || m0 + r >= n) ? (r < 0 ? dpre3_8 : nullptr) : dpre3_8 + (size_t)(m0 + r) * h3; };
        auto b_row = [=] __device__(int c) -> const uint8_t* { return (c < 0 || n0 + c >= h12) ? (c < 0 ? w3_8 : nullptr) : w3_8 + (size_t)(n0 + c) * h3; };
        auto epi = [=] __device__(int, int, float) {};
        auto epiq = [=] __device__(int r0, int c, const float* v) {
            const int nn = n0 + c;
            if (nn >= h12 || m0 + r0 >= NP) return;
            // d A1 / d pre of rows m0 + r0 .. + 3 of column nn: four bf16 = one 8-byte load from the transposed copy the forward left (rows >= n: zeros)
            const uint2 dq = *reinterpret_cast<const uint2*>(dA1T + (size_t)nn * NP + m0 + r0);
            const float da[4] = {__uint_as_float(dq.x << 16), __uint_as_float(dq.x & 0xFFFF0000u), __uint_as_float(dq.y << 16), __uint_as_float(dq.y & 0xFFFF0000u)};
            float g[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) g[x] = m0 + r0 + x < n ? v[x] * da[x] : 0.f;
            *reinterpret_cast<unsigned*>(dpre1T_8 + (size_t)nn * NP + m0 + r0) = ltg_pack4_fp8(g, (float)(1 << FP8_S_G1));
        };
        ltg_sgemm8q<64, 64>(h3, a_row, b_row, 1.f / (float)(1 << (FP8_S_G3 + FP8_S_W)), epi, epiq, s8);
        return;
    }
    bid -= nA;
    if (bid < nB) {
        const int tm = (h12 + 1 + 63) / 64, tn = (h3 + 63) / 64;
        const int z = bid / (tm * tn), t = bid % (tm * tn);
        const int m0 = (t / tn) * 64, n0 = (t % tn) * 64;
        const int kbeg = z * D8_KCHUNK, K = min(NP, kbeg + D8_KCHUNK) - kbeg;
        float* out = slab + (size_t)z * SP;
        const int ow = L.off[4], ob = L.off[5];
        auto a_row = [=] __device__(int r) -> const uint8_t* { return (r < 0 || m0 + r > h12) ? (r < 0 ? A1T_8 : nullptr) : A1T_8 + (size_t)(m0 + r) * NP + kbeg; };
        auto b_row = [=] __device__(int c) -> const uint8_t* { return (c < 0 || n0 + c >= h3) ? (c < 0 ? dpre3T_8 : nullptr) : dpre3T_8 + (size_t)(n0 + c) * NP + kbeg; };
        auto epi = [=] __device__(int r, int c, float g) {
            const int m = m0 + r, nn = n0 + c;
            if (m > h12 || nn >= h3) return;
            if (m < h12) out[ow + (size_t)m * h3 + nn] = g;
            else out[ob + nn] = g;
        };
        ltg_sgemm8q<64, 64>(K, a_row, b_row, 1.f / (float)(1 << (FP8_S_ACT + FP8_S_G3)), epi, LtgNoQuad(), s8);
        return;
    }
    bid -= nB;
    {
        float (*part)[33] = reinterpret_cast<float (*)[33]>(s8);
        const int tc = (h3 + 1 + 31) / 32;
        const int z = bid / tc;
        const int tn = threadIdx.x & 31, tr = threadIdx.x >> 5;
        const int c = (bid % tc) * 32 + tn;  // c == h3 is the bias column
        const int kbeg = z * D8_KCHUNK, kend = min(n, kbeg + D8_KCHUNK);
        float acc = 0.f;
        if (c <= h3) {
#pragma unroll 8
            for (int r = kbeg + tr; r < kend; r += 8) acc += (c < h3 ? A3[(size_t)r * h3 + c] : 1.f) * ds[r];
        }
        part[tr][tn] = acc;
        __syncthreads();
        if (tr == 0 && c <= h3) {
            float g = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) g += part[i][tn];
            slab[(size_t)z * SP + (c < h3 ? L.off[6] + c : L.off[7])] = g;
        }
    }
}

// Backward stage 2: slab[z]: dw1 / db1 = ET_pop . dpre1[:, :h1], dw2 / db2 = ET_niche . dpre1[:, h1:] over the pair rows of chunk z.
__global__ __launch_bounds__(NT) void k8_d_bwd2(int NP, int h0, int h1, int h2, DLayout L, int SP, const uint8_t* __restrict__ ET, const uint8_t* __restrict__ dpre1T_8,
                                                float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) uint8_t s8[2 * (64 + 64) * SG8_LDK];
    const int tm = (h0 + 1 + 63) / 64;
    const int tn1 = (h1 + 63) / 64, tn2 = (h2 + 63) / 64;
    const int per_z = tm * (tn1 + tn2);
    const int z = blockIdx.x / per_z, t = blockIdx.x % per_z;
    // column tile fastest within an XCD's run of blocks would re-read ET; here consecutive blocks share the ROW tile of ET (2 KB x 64
    // rows per chunk) and walk the column tiles of dpre1T
    const int m0 = (t / (tn1 + tn2)) * 64;
    const int tcol = t % (tn1 + tn2);
    const bool br = tcol >= tn1;
    const int n0 = (br ? tcol - tn1 : tcol) * 64;
    const int N = br ? h2 : h1, coff = br ? h1 : 0;
    const int ow = L.off[br ? 2 : 0], ob = L.off[br ? 3 : 1];
    const int kbeg = z * D8_KCHUNK, K = min(NP, kbeg + D8_KCHUNK) - kbeg;
    const uint8_t* E = ET + (size_t)(br ? 1 : 0) * (h0 + 1) * NP;
    float* out = slab + (size_t)z * SP;
    auto a_row = [=] __device__(int r) -> const uint8_t* { return (r < 0 || m0 + r > h0) ? (r < 0 ? E : nullptr) : E + (size_t)(m0 + r) * NP + kbeg; };
    auto b_row = [=] __device__(int c) -> const uint8_t* {
        return (c < 0 || n0 + c >= N) ? (c < 0 ? dpre1T_8 : nullptr) : dpre1T_8 + (size_t)(coff + n0 + c) * NP + kbeg;
    };
    auto epi = [=] __device__(int r, int c, float g) {
        const int m = m0 + r, nn = n0 + c;
        if (m > h0 || nn >= N) return;
        if (m < h0) out[ow + (size_t)m * N + nn] = g;
        else out[ob + nn] = g;
    };
    ltg_sgemm8q<64, 64>(K, a_row, b_row, 1.f / (float)(1 << (FP8_S_EMB + FP8_S_G1)), epi, LtgNoQuad(), s8);
}

// One Adam sweep (train.py:163) over the discriminator with operand-format shadows: g = sum of the chunk slabs.  The three weight
// matrices go tile by tile (64 k-rows x 64 columns): theta / m / v read and written in 16-byte row pieces, the e4m3 of the new theta
// staged in LDS and written TRANSPOSED ([nn][k], k contiguous: the forward layers' operand format) in 16-byte pieces as well --
// and, for w3, once more in its own layout (the backward's operand).  The biases and w4 follow flat.  The last block adds up d_loss.
// Every layer size a multiple of 64.
__global__ __launch_bounds__(NT) void k8_d_adam(int ks, DLayout L, int SP, int h0, int h1, int h2, int h3, int nt1, int nt2, int nt3,
                                                const float* __restrict__ slab, ltg_disc_state st, AdamC ad, int n, const float* __restrict__ lrow,
                                                float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) uint8_t T[64 * 80];
    __shared__ float red[NT / 64];
    const int h12 = h1 + h2;
    int bid = blockIdx.x;
    if (bid < nt1 + nt2 + nt3) {
        int t, N, Kd;
        uint8_t* sht;
        if (bid < nt1) { t = 0; N = h1; Kd = h0; sht = st.w1t_fp8; }
        else if (bid < nt1 + nt2) { bid -= nt1; t = 2; N = h2; Kd = h0; sht = st.w2t_fp8; }
        else { bid -= nt1 + nt2; t = 4; N = h3; Kd = h12; sht = st.w3t_fp8; }
        const int tn = N / 64;
        const int k0 = (bid / tn) * 64, c0 = (bid % tn) * 64;
        const int off = L.off[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = threadIdx.x + NT * e;          // 1024 float4 of the tile: row id / 16, float4 column id % 16
            const int kl = id >> 4, cl = (id & 15) * 4;
            const size_t i = (size_t)(k0 + kl) * N + c0 + cl;
            ltg_f32x4 g = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < ks; ++z) g += *reinterpret_cast<const ltg_f32x4*>(slab + (size_t)z * SP + off + i);
            ltg_f32x4 pp = *reinterpret_cast<const ltg_f32x4*>(st.p[t] + i), mm = *reinterpret_cast<const ltg_f32x4*>(st.m[t] + i),
                      vv = *reinterpret_cast<const ltg_f32x4*>(st.v[t] + i);
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pp[j], mj = mm[j], vj = vv[j];
                adam1(pj, mj, vj, g[j], ad.lr_t, ad);
                pp[j] = pj; mm[j] = mj; vv[j] = vj;
                q[j] = pj;
                T[(cl + j) * 80 + kl] = ltg_f2fp8(pj * (float)(1 << FP8_S_W));
            }
            *reinterpret_cast<ltg_f32x4*>(st.p[t] + i) = pp;
            *reinterpret_cast<ltg_f32x4*>(st.m[t] + i) = mm;
            *reinterpret_cast<ltg_f32x4*>(st.v[t] + i) = vv;
            if (t == 4 && st.w3_fp8) *reinterpret_cast<unsigned*>(st.w3_fp8 + i) = ltg_pack4_fp8(q, (float)(1 << FP8_S_W));
        }
        __syncthreads();
        {
            const int cl = threadIdx.x >> 2, part = threadIdx.x & 3;
            *reinterpret_cast<ltg_u32x4*>(sht + (size_t)(c0 + cl) * Kd + k0 + 16 * part) = *reinterpret_cast<const ltg_u32x4*>(T + cl * 80 + 16 * part);
        }
        return;
    }
    bid -= nt1 + nt2 + nt3;
    // flat part: b1, b2, b3, w4, b4 (tensors 1, 3, 5, 6, 7)
    const int nb = h1 + h2 + h3 + h3 + 1;
    const int nflat = (nb + NT - 1) / NT;
    if (bid < nflat) {
        const int e = bid * NT + threadIdx.x;
        if (e < nb) {
            int t, i;
            if (e < h1) { t = 1; i = e; }
            else if (e < h12) { t = 3; i = e - h1; }
            else if (e < h12 + h3) { t = 5; i = e - h12; }
            else if (e < h12 + 2 * h3) { t = 6; i = e - h12 - h3; }
            else { t = 7; i = 0; }
            float g = 0.f;
            for (int z = 0; z < ks; ++z) g += slab[(size_t)z * SP + L.off[t] + i];
            adam_update(st.p[t], st.m[t], st.v[t], (size_t)i, g, ad);
        }
        return;
    }
    {   // d_loss (train.py:142)
        float s = 0.f;
        if (lrow) for (int i = threadIdx.x; i < n; i += NT) s += lrow[i];
        else for (int z = threadIdx.x; z < ks; z += NT) s += slab[(size_t)z * SP + L.off[8]];
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss_out[0] = s;
    }
}
