// Forward-only discriminator tower as ONE kernel (round 6; discriminator.py:51-55, consumed at train.py:155).
//
// The G step needs only sum_j y_j of the fake tower, and phase G evaluates every tower of the phase ahead (ltg_fake_tower_batched:
// ~93 000 pair rows per sub-epoch of Askubuntu_Sample).  As three launches (fks_d_l1 -> fks_d_l2 -> fk_d_y, ltg_fast.h) the branch
// layers' output A1 [rows][h1 + h2] fp32 went to the workspace and came back once per 64-column tile of the fc layer: 375 MB written and
// 799 MB fetched per sub-epoch (PMC, profiles/r5_pmc_traffic.json) for an activation nobody else reads.  Here a workgroup owns 64 pair
// rows from the id lists to y:
//
//   gather E[pop], E[niche] -> both branch layers -> A1 [64][h1 + h2] fp32 in LDS -> fc layer straight from LDS, w3 streamed from the
//   L2 -> dropout(tanh) . w4 reduced per row -> sigmoid -> y[row]
//
// Arithmetic: fp32-accurate on the bf16 matrix pipe (ltg_rgemm.h: x = hi + mid + lo exactly in bf16 terms, six cross terms per product,
// fp32 accumulation; SPL = 4: four terms, opt-in).  The WEIGHTS are split once per launch sequence by fkt_split_weights into MFMA
// fragment order -- [16-column tile][32-deep k pair][term][lane] x 16 bytes, so a wave's B fragment is ONE coalesced 1-KiB request and
// costs no vector arithmetic in the loop; activations (gathered embedding rows, A1 out of LDS) are split by the wave that multiplies them.
// k -> (lane, slot) map of v_mfma_f32_16x16x32_bf16 fragments: lane (r, q), slot s  <->  k = 32 p + 8 q + s, the same for both operands.
//
// 512 threads: layer 1  wave = (32-row half, branch, column half)  [waves 0-3 popular -> h1 columns, 4-7 niche -> h2 columns: each SIMD
//                       hosts one wave of either branch]
//              layer 2  wave = (32-row half, one of four column groups of NTW 16-column tiles)
// Epilogue per element: tanh from v_exp_f32 / v_rcp_f32 (polynomial below |x| = 0.04), the dropout draw of ltg_rng.h with the row part
// and the column part of its counter hoisted (same bits), 1 / keep as a multiplication.
#pragma once

// (FT_BM, FT_NT, FT_KP1_MAX, ft_lda, ft_lds_bytes, ft_wsp_triples, ft_wsp_bytes: ltg_kernels.hip, in front of the workspace layout that uses them)
typedef unsigned ltg_ft_u32x4 __attribute__((ext_vector_type(4)));

// one wave per (matrix, 16-column tile, k pair): w1 | w2 | w3 in that order
__global__ __launch_bounds__(NT) void fkt_split_weights(int h0, int h1, int h2, int h3, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ w3, ltg_ft_u32x4* __restrict__ wsp) {
    const int wv = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int h12 = h1 + h2, KP1 = (h0 + 31) / 32, KP3 = (h12 + 31) / 32;
    const int n1 = ((h1 + 15) / 16) * KP1, n2 = ((h2 + 15) / 16) * KP1, n3 = ((h3 + 15) / 16) * KP3;
    if (wv >= n1 + n2 + n3) return;
    const float* W = wv < n1 ? w1 : (wv < n1 + n2 ? w2 : w3);
    const int K = wv < n1 + n2 ? h0 : h12, N = wv < n1 ? h1 : (wv < n1 + n2 ? h2 : h3);
    const int f = wv < n1 ? wv : (wv < n1 + n2 ? wv - n1 : wv - n1 - n2);
    const int KP = (K + 31) / 32, tile = f / KP, p = f % KP;
    const int n = 16 * tile + r;
    ltg_f32x4 x0, x1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k0 = 32 * p + 8 * q + s, k1 = k0 + 4;
        x0[s] = (k0 < K && n < N) ? W[(size_t)min(k0, K - 1) * N + min(n, N - 1)] : 0.f;
        x1[s] = (k1 < K && n < N) ? W[(size_t)min(k1, K - 1) * N + min(n, N - 1)] : 0.f;
    }
    const LtgSplit s0 = ltg_split_bf16<6>(x0), s1 = ltg_split_bf16<6>(x1);
    ltg_ft_u32x4* out = wsp + (size_t)wv * 3 * 64 + lane;
    out[0] = ltg_ft_u32x4{s0.hi[0], s0.hi[1], s1.hi[0], s1.hi[1]};
    out[64] = ltg_ft_u32x4{s0.mid[0], s0.mid[1], s1.mid[0], s1.mid[1]};
    out[128] = ltg_ft_u32x4{s0.lo[0], s0.lo[1], s1.lo[0], s1.lo[1]};
}

// verification helper (ltg_debug_split): the three bf16 terms of every input value, as the GEMM loaders form them -- out[3 i + t] = term t of in[i] as fp32
__global__ __launch_bounds__(NT) void k_debug_split(int n4, const ltg_f32x4* __restrict__ in, float* __restrict__ out) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n4; i += gridDim.x * NT) {
        const LtgSplit s = ltg_split_bf16<6>(in[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned sh = (e & 1) ? 0u : 16u;
            out[3 * (4 * i + e) + 0] = __uint_as_float(e & 1 ? (s.hi[e >> 1] & 0xFFFF0000u) : (s.hi[e >> 1] << sh));
            out[3 * (4 * i + e) + 1] = __uint_as_float(e & 1 ? (s.mid[e >> 1] & 0xFFFF0000u) : (s.mid[e >> 1] << sh));
            out[3 * (4 * i + e) + 2] = __uint_as_float(e & 1 ? (s.lo[e >> 1] & 0xFFFF0000u) : (s.lo[e >> 1] << sh));
        }
    }
}

// tanh: 1 - 2 / (1 + e^(2x)) through v_exp_f32 / v_rcp_f32 (absolute error ~1e-7; saturates cleanly), the odd polynomial where that
// form cancels (|x| < 0.04: next term 17/315 x^7 < 1e-11)
__device__ __forceinline__ float ft_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // e^(2x)
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
    const float x2 = x * x;
    const float small = x * fmaf(x2, fmaf(x2, 0.13333333333f, -0.33333333333f), 1.f);
    return fabsf(x) < 0.04f ? small : big;
}

// The dropout draw of DropView::keep with the counter hash of ltg_rng.h taken apart: z0 = [(seed ^ stream K1) + step K2 + (row_local width + 1) K3]
// + column K3 -- the bracket once per row (ft_drop_rows), column K3 once per column, one 64-bit add per element; the two mixing rounds per
// element.  keep <=> (z >> 40) 2^-24 < kp <=> (z >> 40) < ceil(kp 2^24) (both sides integers below 2^24 + 1; only the top 24 bits of the last
// product are needed, and the final z ^= z >> 31 does not reach them).  Same bits as ltg_rng_keep.
// The row parts of the 8 rows a lane finishes (C layout: rows 4 q + x of its two 16-row tiles): the batched tower's lookups (segment -> counter,
// first row) are requested ONCE at the top of the kernel and serve all three dropout streams.
struct FtRows {
    uint64_t st[2][4];      // the row's dropout counter
    uint32_t rl[2][4];      // the row's number inside its pair batch
};
__device__ __forceinline__ FtRows ft_rows(const DropView& dv, int mrow0, int q, int n, uint64_t step) {
    FtRows R;
    if (dv.seg_of) {        // (uniform: ONE branch around the lookups)
        int sg[2][4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int x = 0; x < 4; ++x) sg[tm][x] = dv.seg_of[min(mrow0 + 16 * tm + 4 * q + x, n - 1)];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                R.st[tm][x] = dv.seg_step[sg[tm][x]];
                R.rl[tm][x] = (uint32_t)(min(mrow0 + 16 * tm + 4 * q + x, n - 1) - dv.seg_row0[sg[tm][x]]);
            }
    } else {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                R.st[tm][x] = step;
                R.rl[tm][x] = (uint32_t)(min(mrow0 + 16 * tm + 4 * q + x, n - 1) + dv.row0);
            }
    }
    return R;
}
__device__ __forceinline__ void ft_drop_rows(uint64_t (&zrow)[2][4], const FtRows& R, int width, uint64_t seed, uint32_t stream) {
    const uint64_t k0 = seed ^ ((uint64_t)stream * 0xD6E8FEB86659FD93ull);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int x = 0; x < 4; ++x)
            zrow[tm][x] = k0 + R.st[tm][x] * 0x94D049BB133111EBull + ((uint64_t)R.rl[tm][x] * (uint64_t)width + 1ull) * 0x9E3779B97F4A7C15ull;
}
__device__ __forceinline__ bool ft_drop_keep(uint64_t zrow, uint64_t zcol, unsigned thr) {
    uint64_t z = zrow + zcol;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)(z >> 40) < thr;
}

template <int SPL>
__device__ __forceinline__ void ft_mfma_terms(ltg_f32x4& acc, const ltg_ft_u32x4 (&a)[3], const ltg_ft_u32x4 (&b)[3]) {
    auto mm = [&] __device__(const ltg_ft_u32x4& x, const ltg_ft_u32x4& y) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ltg_bf16x8, x), __builtin_bit_cast(ltg_bf16x8, y), acc, 0, 0, 0);
    };
    if constexpr (SPL == 6) {      // the small terms first
        mm(a[0], b[2]);
        mm(a[2], b[0]);
    }
    mm(a[1], b[1]);
    mm(a[0], b[1]);
    mm(a[1], b[0]);
    mm(a[0], b[0]);
}
// two fp32 quads (k = 8 q .. 8 q + 7 of one row) -> the three bf16 terms of the fragment
template <int SPL>
__device__ __forceinline__ void ft_split_frag(ltg_f32x4 x0, ltg_f32x4 x1, ltg_ft_u32x4 (&out)[3]) {
    const LtgSplit s0 = ltg_split_bf16<SPL>(x0), s1 = ltg_split_bf16<SPL>(x1);
    out[0] = ltg_ft_u32x4{s0.hi[0], s0.hi[1], s1.hi[0], s1.hi[1]};
    out[1] = ltg_ft_u32x4{s0.mid[0], s0.mid[1], s1.mid[0], s1.mid[1]};
    out[2] = ltg_ft_u32x4{s0.lo[0], s0.lo[1], s1.lo[0], s1.lo[1]};
}

// NTW: 16-column tiles of the fc layer per wave (four wave columns): 5 for h3 <= 320.  INJ: the caller injects the dropout masks (parity runs):
// DropView::keep per element instead of the hoisted counter hash.
template <int SPL, int NTW, bool INJ>
__global__ __launch_bounds__(FT_NT) void fkt_d_tower(PairView pv, int h0, int h1, int h2, int h3, const float* __restrict__ emb,
                                                     const ltg_ft_u32x4* __restrict__ wsp, const float* __restrict__ b1, const float* __restrict__ b2,
                                                     const float* __restrict__ b3, const float* __restrict__ w4, const float* __restrict__ b4,
                                                     DropView dA, DropView dB, DropView dC, float keep, uint64_t seed, uint64_t step,
                                                     float* __restrict__ y) {
    static_assert(SPL == 6 || SPL == 4, "the split arithmetic (d_arith = fp32 keeps the three-launch tower on the fp32 matrix pipe)");
    constexpr int NTERM = SPL == 6 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) float ft_lds[];
    const int n = pv.nr + pv.nf, h12 = h1 + h2, LDA = ((h12 + 31) & ~31) + 4;
    float* A1s = ft_lds;                       // [64][LDA]
    float* red = ft_lds + FT_BM * LDA;         // [4][64]: the wave columns' shares of A3 . w4
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * FT_BM;
    const int KP1 = (h0 + 31) / 32, KP3 = (h12 + 31) / 32;
    const float inv_keep = 1.f / keep;
    const unsigned thr = keep >= 1.f ? 0x1000000u : (unsigned)ceilf(keep * 16777216.f);
    const int mh = w & 1;
    // the K padding of A1 (columns h12 .. 32 KP3 - 1) is read by the last k pair of layer 2
    for (int e = tid; e < FT_BM * (32 * KP3 - h12); e += FT_NT) A1s[(e / (32 * KP3 - h12)) * LDA + h12 + e % (32 * KP3 - h12)] = 0.f;

    // the dropout draws' row lookups: requested first, consumed by the epilogues (their round trips overlap the gather's)
    FtRows R;
    if constexpr (!INJ) R = ft_rows(dA, m0 + 32 * mh, q, n, step);
    // layer 2's first B fragments do not depend on layer 1 either
    const int ng = w >> 1;
    const int nt3 = (h3 + 15) / 16;
    const ltg_ft_u32x4* wb3 = wsp + (size_t)(((h1 + 15) / 16) + ((h2 + 15) / 16)) * KP1 * 3 * 64 + lane;
    int tl[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) tl[j] = min(ng * NTW + j, nt3 - 1);       // (tiles beyond the last one recompute it; their results are dropped)

    // ---- layer 1: A1 = dropout(tanh(E . w + b)), this wave's 32 rows x its half of one branch's columns
    {
        const bool br = w >= 4;
        const int nh = (w >> 1) & 1;
        const int Nb = br ? h2 : h1, coff = br ? h1 : 0;
        const int ntb = (Nb + 15) / 16, half = (ntb + 1) / 2;
        const int t0 = nh * half, t1 = min(ntb, t0 + half);
        const float* bias = br ? b2 : b1;
        const DropView& dv = br ? dB : dA;
        const uint32_t stream = br ? LTG_STREAM_D_DROP_B : LTG_STREAM_D_DROP_A;
        const ltg_ft_u32x4* wb = wsp + (size_t)(br ? ((h1 + 15) / 16) * KP1 : 0) * 3 * 64 + lane;
        // ONE B fragment set, refilled for the next tile as soon as this tile's MFMAs are issued: in flight under the epilogue
        ltg_ft_u32x4 bf[FT_KP1_MAX][3];
        auto load_b = [&] __device__(int t) {
            const int tc = min(t, t1 - 1);
#pragma unroll
            for (int p = 0; p < FT_KP1_MAX; ++p)
#pragma unroll
                for (int tt = 0; tt < NTERM; ++tt) bf[p][tt] = wb[((size_t)tc * KP1 + min(p, KP1 - 1)) * 3 * 64 + tt * 64];
        };
        load_b(t0);
        // A fragments: the embedding rows of this lane's two operand rows, split, all k pairs
        ltg_ft_u32x4 af[2][FT_KP1_MAX][3];
        {
            ltg_f32x4 raw[2][FT_KP1_MAX][2];
            bool rok[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int m = m0 + 32 * mh + 16 * tm + r, mc = min(m, n - 1);
                const int id = br ? pv.nic(mc) : pv.pop(mc);
                rok[tm] = m < n && id >= 0;
                const float* erow = emb + (size_t)max(id, 0) * h0;
#pragma unroll
                for (int p = 0; p < FT_KP1_MAX; ++p)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) raw[tm][p][hh] = ltg_ld4(erow + min(32 * p + 8 * q + 4 * hh, h0 - 4));
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int p = 0; p < FT_KP1_MAX; ++p) {
                    ltg_f32x4 x[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const bool ok = rok[tm] && 32 * p + 8 * q + 4 * hh < h0;       // h0 % 4 == 0: a quad is inside or outside
                        x[hh] = ok ? raw[tm][p][hh] : ltg_f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    ft_split_frag<SPL>(x[0], x[1], af[tm][p]);
                }
        }
        uint64_t zrow[2][4];
        if constexpr (!INJ) ft_drop_rows(zrow, R, Nb, seed, stream);
        auto product = [&] __device__(ltg_f32x4 (&acc)[2]) {       // (k pairs beyond h0 multiply zeroed A fragments: no branch)
            acc[0] = acc[1] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < FT_KP1_MAX; ++p)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) ft_mfma_terms<SPL>(acc[tm], af[tm][p], bf[p]);
        };
        auto finish = [&] __device__(const ltg_f32x4 (&acc)[2], int t) {
            const int col = 16 * t + r, cc = min(col, Nb - 1);
            const float bv = bias[cc];
            const uint64_t zcol = (uint64_t)cc * 0x9E3779B97F4A7C15ull;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int row = 32 * mh + 16 * tm + 4 * q + x, m = m0 + row;
                    const float tv = ft_tanh(acc[tm][x] + bv);
                    bool kp;
                    if constexpr (INJ) kp = dv.keep(min(m, n - 1), cc, Nb, seed, stream, step, keep);
                    else kp = ft_drop_keep(zrow[tm][x], zcol, thr);
                    if (col < Nb) A1s[row * LDA + coff + col] = (kp && m < n) ? tv * inv_keep : 0.f;
                }
        };
        // tile t + 1's MFMAs are issued in front of tile t's epilogue (the matrix pipe works under the epilogue's vector instructions), the B
        // fragments of tile t + 2 are requested behind them
        if (t0 < t1) {
            ltg_f32x4 accA[2], accB[2];
            product(accA);
            load_b(t0 + 1);
            for (int t = t0; t < t1; t += 2) {
                product(accB);
                load_b(t + 2);
                finish(accA, t);
                if (t + 1 < t1) {
                    product(accA);
                    load_b(t + 3);
                    finish(accB, t + 1);
                }
            }
        }
    }
    // layer 2's B fragments of the first TWO k pairs: in flight across the barrier
    ltg_ft_u32x4 bfe[NTW][3], bfo[NTW][3];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int tt = 0; tt < NTERM; ++tt) {
            bfe[j][tt] = wb3[((size_t)tl[j] * KP3) * 3 * 64 + tt * 64];
            bfo[j][tt] = wb3[((size_t)tl[j] * KP3 + min(1, KP3 - 1)) * 3 * 64 + tt * 64];
        }
    __syncthreads();

    // ---- layer 2: A3 = dropout(tanh(A1 . w3 + b3)), reduced against w4 on the way out
    {
        ltg_f32x4 acc[2][NTW];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[tm][j] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) arow[tm] = A1s + (32 * mh + 16 * tm + r) * LDA + 8 * q;
        // TWO fragment sets (even / odd k pairs), each refilled tile by tile for pair p + 2 as soon as the tile's MFMAs of pair p are issued: a
        // request has two k pairs of MFMAs (~2 000 cycles) to land
        auto block = [&] __device__(ltg_ft_u32x4 (&bf)[NTW][3], int p) {
            ltg_ft_u32x4 af[2][3];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const ltg_f32x4 x0 = *reinterpret_cast<const ltg_f32x4*>(arow[tm] + 32 * p), x1 = *reinterpret_cast<const ltg_f32x4*>(arow[tm] + 32 * p + 4);
                ft_split_frag<SPL>(x0, x1, af[tm]);
            }
            const int pn = min(p + 2, KP3 - 1);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) ft_mfma_terms<SPL>(acc[tm][j], af[tm], bf[j]);
#pragma unroll
                for (int tt = 0; tt < NTERM; ++tt) bf[j][tt] = wb3[((size_t)tl[j] * KP3 + pn) * 3 * 64 + tt * 64];
            }
        };
        for (int p = 0; p < KP3; p += 2) {
            block(bfe, p);
            if (p + 1 < KP3) block(bfo, p + 1);
        }
        // epilogue: this wave column's share of A3[row] . w4 for its 32 rows
        uint64_t zrow[2][4];
        if constexpr (!INJ) ft_drop_rows(zrow, R, h3, seed, LTG_STREAM_D_DROP_C);
        float pd[2][4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int x = 0; x < 4; ++x) pd[tm][x] = 0.f;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = 16 * (ng * NTW + j) + r, cc = min(col, h3 - 1);
            const bool cok = ng * NTW + j < nt3 && col < h3;
            const float bv = b3[cc], w4v = w4[cc], wv = cok ? w4v : 0.f;
            const uint64_t zcol = (uint64_t)cc * 0x9E3779B97F4A7C15ull;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int m = m0 + 32 * mh + 16 * tm + 4 * q + x;
                    const float tv = ft_tanh(acc[tm][j][x] + bv);
                    bool kp;
                    if constexpr (INJ) kp = dC.keep(min(m, n - 1), cc, h3, seed, LTG_STREAM_D_DROP_C, step, keep);
                    else kp = ft_drop_keep(zrow[tm][x], zcol, thr);
                    pd[tm][x] += kp ? tv * inv_keep * wv : 0.f;
                }
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float v = pd[tm][x];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);      // the 16 lanes r of a quad group
                if (r == 0) red[ng * FT_BM + 32 * mh + 16 * tm + 4 * q + x] = v;
            }
    }
    __syncthreads();
    if (tid < FT_BM && m0 + tid < n) {
        const int m = m0 + tid;
        const float s = b4[0] + red[tid] + red[FT_BM + tid] + red[2 * FT_BM + tid] + red[3 * FT_BM + tid];     // fixed order: reproducible
        y[m] = pv.valid(m) ? 1.f / (1.f + expf(-s)) : 0.f;
    }
}
