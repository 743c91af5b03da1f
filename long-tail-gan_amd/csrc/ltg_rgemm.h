// Register-resident MFMA GEMM block for the latency-bound shapes of the path (gfx950, wave64).
//
// The middle layers of the generator (100 x 600 x 400 ...), the discriminator layers (1 800 x 400 x 300 ...) and every
// weight-gradient product of a 100-user batch are far too small to be bandwidth- or MFMA-bound: a launch lasts as long
// as its longest chain of DEPENDENT memory round trips.  The LDS-staged template of ltg_gemm.h pays one round trip per
// K tile (load -> barrier -> LDS -> MFMA, 3-5 times per workgroup); this block pays ONE: every operand element a wave
// needs is requested up front, straight into registers in MFMA fragment order, the MFMAs run as the data lands, and the
// only barrier is the final meeting of the K slices.
//
//   C[m][n] = sum_k A(m, k) * B(k, n),   v_mfma_f32_16x16x4_f32 (exact fp32 fma chain; operands that the decoder GEMMs
//   round to bf16 are rounded by their loaders: bf16 x bf16 products are exact in fp32, so this equals the bf16 MFMA with
//   fp32 accumulation up to summation order)
//
// Geometry: a 256-thread workgroup = 4 waves arranged WM x WN x WK; a wave owns (16 TM) x (16 TN) outputs and one of WK
// slices of K; the WK partial tiles meet in LDS and the epilogue walks the (16 TM WM) x (16 TN WN) tile row-major.
// K is cut into blocks of 16: lane (r = l & 15, q = l >> 4) fetches k = 16 jb + 4 q .. + 3 of its row (A) / column (B),
// and step j = 0..3 of the block feeds element j of both fragments to one MFMA -- the k -> (lane, step) map is a
// permutation both operands share, so 16 bytes per lane and access serve four MFMAs.
//
// Operand functors come in pairs (no branches; the block clamps indices into range, mirrors rows / columns beyond M / N onto
// the last one -- the epilogue skips them -- and zeroes the A fragment of k blocks beyond K):
//   a_ld(i, m, k) -> RAW {A(m,k), A(m,k+1), A(m,k+2), A(m,k+3)} (or a struct of several such requests that a_xf folds into one
//   operand, e.g. a value and the activation its derivative needs): memory requests only, no arithmetic on what they return
//   a_xf(raw, i, m, k) -> the operand values (rounding, scaling, masks, ones-augmentation)           same for b_ld(i, k, n) / b_xf
//   with k % 4 == 0 and i = the index of the k block within the pass.  The split is what makes "one round trip" real:
//   phase 1 of a pass holds nothing but address arithmetic and loads, a scheduling barrier closes it, and only then do the
//   transforms and MFMAs consume the registers.  (With the arithmetic next to its load the compiler waits after every
//   load -- vmcnt(0) fourteen times per pass in the first version of the discriminator's second layer.)
//   When K % 4 != 0 (K = pair rows / batch rows) the functors clamp the addresses of elements with k + j >= K and a_xf
//   zeroes them.
//   epi(e, m, n, value, in_range) (e = 0, 1, ...: the call's index) is called by every thread the same number of times (in_range = m < M && n < N), so it may
//   use wave shuffles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ltg_gemm.h"

__device__ __forceinline__ ltg_f32x4 ltg_ld4(const float* __restrict__ p) { return *reinterpret_cast<const ltg_f32x4*>(p); }
// four elements of a strided column: p[0], p[s], p[2s], p[3s]
__device__ __forceinline__ ltg_f32x4 ltg_ld4s(const float* __restrict__ p, size_t s) { return ltg_f32x4{p[0], p[s], p[2 * s], p[3 * s]}; }
__device__ __forceinline__ float ltg_bf16r(float x) { return __uint_as_float((unsigned)ltg_f2bf(x) << 16); }
__device__ __forceinline__ ltg_f32x4 ltg_bf16r4(ltg_f32x4 v) { return ltg_f32x4{ltg_bf16r(v[0]), ltg_bf16r(v[1]), ltg_bf16r(v[2]), ltg_bf16r(v[3])}; }

// MEASUREMENT BUILD ONLY (-DLTG_STAMP=<id>): the workgroups of ONE kernel (the one that instantiates the block with SID == LTG_STAMP) leave
// 100-MHz wall-clock stamps of their phases in ltg_stamp_buf[workgroup][8] (scripts/stamp_probe.py reads them through ltg_debug_stamps):
// 0 kernel entry (the kernel's own first statement), 1 operand requests issued, 2 mid hook done, 3 last MFMA issued, 4 K slices met in LDS,
// 5 epilogue done (stores issued), 6 HW_ID register, 7 unused
#ifdef LTG_STAMP
__device__ unsigned long long ltg_stamp_buf[16384 * 8];
__device__ __forceinline__ void ltg_stamp(int slot) {
    if (threadIdx.x == 0) {
        const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 16384u)
            ltg_stamp_buf[wg * 8 + slot] = slot == 6 ? ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                                        ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32))   // HW_ID | XCC_ID << 32
                                                     : wall_clock64();
    }
}
#define LTG_STAMP_AT(SID, slot) do { if constexpr ((SID) == LTG_STAMP) ltg_stamp(slot); } while (0)
#else
#define LTG_STAMP_AT(SID, slot) do { } while (0)
#endif

// fp32 operands on the bf16 matrix pipe without giving up fp32 accuracy (round 6; discriminator GEMMs, ltg_config.d_arith): x = hi + mid + lo EXACTLY with
// hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (RNE: each residual is exact in fp32 and 8 bits shorter, |mid| <= 2^-9 |x|,
// |lo| <= 2^-18 |x|), every bf16 x bf16 product is exact in fp32, and the accumulation is the matrix pipe's fp32.  Six cross terms
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi) leave out 2 * 2^-27 |a b| per product -- below the fp32 rounding of the product itself; four
// terms (no lo) leave out 2^-17.  Two elements per dword as v_mfma_f32_16x16x32_bf16 wants them: v_cvt_pk_bf16_f32 + v_pk_add_f32, 18 vector
// instructions per four elements (SPL = 6) / 10 (SPL = 4).
struct LtgSplit {
    unsigned hi[2], mid[2], lo[2];
};
template <int SPL>
__device__ __forceinline__ LtgSplit ltg_split_bf16(ltg_f32x4 v) {
    typedef float ltg_rg_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 ltg_rg_bf16x2 __attribute__((ext_vector_type(2)));
    LtgSplit s;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const ltg_rg_f32x2 x = {v[2 * h], v[2 * h + 1]};
        const ltg_rg_bf16x2 hb = __builtin_convertvector(x, ltg_rg_bf16x2);
        const unsigned hu = __builtin_bit_cast(unsigned, hb);
        // (Measured and not kept, round 6: the residual as v_dot2c_f32_bf16 -- x + hi.lo * -1 + hi.hi * 0, one instruction per element instead of
        // the unpacking shift / and + half a v_pk_add_f32.  No gain -- D phase 47.97 -> 47.78, G phase 62.88 -> 63.01 ms -- and as hipcc 7.2 compiles
        // the builtin here (the packed selector becomes an inline constant of the instruction) the terms come out WRONG: every D-step parity case
        // and test_bf16_split_is_exact fail on that build.)
        const ltg_rg_f32x2 r1 = x - ltg_rg_f32x2{__uint_as_float(hu << 16), __uint_as_float(hu & 0xFFFF0000u)};
        const ltg_rg_bf16x2 mb = __builtin_convertvector(r1, ltg_rg_bf16x2);
        const unsigned mu = __builtin_bit_cast(unsigned, mb);
        s.hi[h] = hu;
        s.mid[h] = mu;
        s.lo[h] = 0u;
        if constexpr (SPL == 6) {
            const ltg_rg_f32x2 r2 = r1 - ltg_rg_f32x2{__uint_as_float(mu << 16), __uint_as_float(mu & 0xFFFF0000u)};
            s.lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, ltg_rg_bf16x2));
        }
    }
    return s;
}

template <int TM, int TN, int WM, int WN, int WK>
struct LtgRg {
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    static constexpr int LDC = BN + 4;
    static constexpr int LDS_FLOATS = WK * BM * LDC;
    // row of tile tm this lane loads for operand A (unclamped)
    static __device__ __forceinline__ int row(int m0, int tm) {
        const int w = threadIdx.x >> 6, wm = w / (WK * WN);
        return m0 + (wm * TM + tm) * 16 + (threadIdx.x & 15);
    }
    // first k of block i of this lane's K slice (single pass), clamped like the block clamps it
    static __device__ __forceinline__ int kc(int K, int i) {
        const int w = threadIdx.x >> 6, wk = w % WK, q = (threadIdx.x & 63) >> 4;
        const int per = (((K + 15) >> 4) + WK - 1) / WK;
        return min(16 * (wk * per + i) + 4 * q, K >= 4 ? ((K - 1) & ~3) : 0);
    }
    static __device__ __forceinline__ int col(int n0, int tn) {
        const int w = threadIdx.x >> 6, wn = (w / WK) % WN;
        return n0 + (wn * TN + tn) * 16 + (threadIdx.x & 15);
    }
};

// NBLK: 16-deep k blocks a wave keeps in flight at once (registers: NBLK * (TM + TN) * 4).  K slices longer than NBLK
// blocks take several passes (one memory round trip each).
// Product phase: leaves the WK partial tiles in LDS ([WK][BM][LDC]) behind a barrier.
struct LtgNoMid {
    __device__ __forceinline__ void operator()() const {}
};
struct LtgXfId {
    __device__ __forceinline__ ltg_f32x4 operator()(ltg_f32x4 v, int, int, int) const { return v; }
};

// mid(): called once, after the requests of the first pass have been issued and before anything consumes them -- the place
// for work that needs an earlier load of the caller's (e.g. row factors into LDS + a barrier) without costing a round trip.
// SPL = 0: v_mfma_f32_16x16x4_f32 (the exact fp32 fma chain).  SPL = 6 / 4: the same product as bf16 cross terms of the split operands
// (ltg_split_bf16) on v_mfma_f32_16x16x32_bf16 -- two 16-deep blocks share an MFMA: slots 0..3 of a lane's fragment are block i's k = 4 q + j,
// slots 4..7 block i + 1's, for both operands, so the instruction adds up 32 k of one cross term; six (four) MFMAs of 16 cycles per block pair
// and output tile instead of eight of 32.
template <int TM, int TN, int WM, int WN, int WK, int NBLK, bool PEEL = false, int SID = 0, int SPL = 0, class ALD, class AXF, class BLD, class BXF, class MID = LtgNoMid>
__device__ __forceinline__ void ltg_rgemm_product(int M, int N, int K, int m0, int n0, ALD a_ld, AXF a_xf, BLD b_ld, BXF b_xf, float* __restrict__ lds,
                                                  MID mid = MID()) {
    static_assert(SPL == 0 || SPL == 4 || SPL == 6, "fp32 MFMA, or the four- / six-term bf16 split");
    static_assert(WM * WN * WK == 4 || WM * WN * WK == 8, "4 or 8 waves per workgroup");
    typedef LtgRg<TM, TN, WM, WN, WK> G;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int nblk = (K + 15) >> 4;
    const int per = (nblk + WK - 1) / WK;
    const int Kc = K >= 4 ? ((K - 1) & ~3) : 0;   // start of the last (possibly partial) group of 4
    int am[TM], bn[TN];
    bool aok[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 16 + r;
        aok[tm] = m < M;
        am[tm] = min(m, M - 1);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bn[tn] = min(n0 + (wn * TN + tn) * 16 + r, N - 1);
    ltg_f32x4 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    // (round 5, PEEL: the first pass peeled off the loop.  As the header of a loop, the pass begins with the waits the back edge needs -- the
    // registers the requests land in are the targets of the previous pass's loads -- and those waits also drain whatever the CALLER has in flight on
    // first entry (pair ids, tile partials, activations of the epilogue).  Measured same-box on Askubuntu_Sample: peeled everywhere D phase 54.85 ->
    // 53.6 ms per epoch but G phase 69.1 -> 70.5 (twice the code for launches that last 5-9 us): the discriminator's kernels peel, the
    // generator's do not.)
    auto pass = [&] __device__(const int base) {
        decltype(a_ld(0, 0, 0)) ra[NBLK][TM];   // RAW operand requests: a float4, or a small struct of them (a_xf folds it)
        decltype(b_ld(0, 0, 0)) rb[NBLK][TN];
        // phase 1: requests only
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const int kc = min(16 * (wk * per + base + i) + 4 * q, Kc);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) ra[i][tm] = a_ld(i, am[tm], kc);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) rb[i][tn] = b_ld(i, kc, bn[tn]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (base == 0) {
            LTG_STAMP_AT(SID, 1);
            mid();
            LTG_STAMP_AT(SID, 2);
        }
        if constexpr (SPL != 0) {
            // phase 2, split form: pairs of blocks; a block of the pair that is out of range contributes zeros (its A fragment is zeroed)
            typedef unsigned ltg_rg_u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < NBLK; i += 2) {
                if (base + i >= per || 16 * (wk * per + base + i) >= K) continue;   // wave-uniform: nothing of this pair is in range
                LtgSplit sa[TM][2], sb[TN][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (i + h < NBLK) {
                        const int k = 16 * (wk * per + base + i + h) + 4 * q;
                        const bool ok = base + i + h < per && k < K;
                        const int kc = min(k, Kc);
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            const ltg_f32x4 v = a_xf(ra[i + h][tm], i + h, am[tm], kc);
                            const bool o = ok && aok[tm];
                            sa[tm][h] = ltg_split_bf16<SPL>(ltg_f32x4{o ? v[0] : 0.f, o ? v[1] : 0.f, o ? v[2] : 0.f, o ? v[3] : 0.f});
                        }
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) sb[tn][h] = ltg_split_bf16<SPL>(b_xf(rb[i + h][tn], i + h, kc, bn[tn]));
                    } else {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) sa[tm][h] = LtgSplit{{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) sb[tn][h] = LtgSplit{{0u, 0u}, {0u, 0u}, {0u, 0u}};
                    }
                }
                ltg_bf16x8 ah[TM], am_[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    ah[tm] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sa[tm][0].hi[0], sa[tm][0].hi[1], sa[tm][1].hi[0], sa[tm][1].hi[1]});
                    am_[tm] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sa[tm][0].mid[0], sa[tm][0].mid[1], sa[tm][1].mid[0], sa[tm][1].mid[1]});
                    al[tm] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sa[tm][0].lo[0], sa[tm][0].lo[1], sa[tm][1].lo[0], sa[tm][1].lo[1]});
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    bh[tn] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sb[tn][0].hi[0], sb[tn][0].hi[1], sb[tn][1].hi[0], sb[tn][1].hi[1]});
                    bm[tn] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sb[tn][0].mid[0], sb[tn][0].mid[1], sb[tn][1].mid[0], sb[tn][1].mid[1]});
                    bl[tn] = __builtin_bit_cast(ltg_bf16x8, ltg_rg_u32x4{sb[tn][0].lo[0], sb[tn][0].lo[1], sb[tn][1].lo[0], sb[tn][1].lo[1]});
                }
                // the small terms first
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        if constexpr (SPL == 6) {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        }
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_[tm], bm[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bm[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }
            return;
        }
        // phase 2: transforms and MFMAs, block by block as the data lands
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const int k = 16 * (wk * per + base + i) + 4 * q;
            const bool ok = base + i < per && k < K;
            const int kc = min(k, Kc);
            if (base + i >= per || 16 * (wk * per + base + i) >= K) continue;   // wave-uniform: nothing of this block is in range
            ltg_f32x4 av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const ltg_f32x4 v = a_xf(ra[i][tm], i, am[tm], kc);
                const bool o = ok && aok[tm];
                av[tm] = ltg_f32x4{o ? v[0] : 0.f, o ? v[1] : 0.f, o ? v[2] : 0.f, o ? v[3] : 0.f};
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[tn] = b_xf(rb[i][tn], i, kc, bn[tn]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm][j], bv[tn][j], acc[tm][tn], 0, 0, 0);
        }
    };
    if constexpr (PEEL) {
        pass(0);
        for (int base = NBLK; base < per; base += NBLK) pass(base);
    } else {
        for (int base = 0; base < per; base += NBLK) pass(base);
    }
    LTG_STAMP_AT(SID, 3);
    float* mine = lds + wk * (G::BM * G::LDC);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int x = 0; x < 4; ++x)
                mine[((wm * TM + tm) * 16 + 4 * q + x) * G::LDC + (wn * TN + tn) * 16 + r] = acc[tm][tn][x];
    __syncthreads();
    LTG_STAMP_AT(SID, 4);
}

// The same product on the bf16 matrix pipe, for the three decoder GEMMs whose operands the loaders round to bf16 anyway (round 5): the
// products bf16 x bf16 are exact in fp32 either way, so v_mfma_f32_16x16x32_bf16 on the PACKED operands adds up the same terms as
// v_mfma_f32_16x16x4_f32 on the rounded fp32 values (in another order) -- at 16 cycles per 32-deep block instead of 8 x 32.  At Askubuntu's
// sizes the fp32 form kept the matrix pipe busy for 1-1.2 us of each of these 8-us kernels.
// K is cut into blocks of 32: lane (r, q) fetches k = 32 jb + 8 q .. + 7 of its row / column as TWO requests of four (the functors' unit).
// a_xf / b_xf must return values that are exactly representable in bf16 (ltg_bf16r4) or zero; NBLK counts 32-deep blocks.
template <int TM, int TN, int WM, int WN, int WK, int NBLK, bool PEEL = false, class ALD, class AXF, class BLD, class BXF, class MID = LtgNoMid>
__device__ __forceinline__ void ltg_rgemm_product_bf16(int M, int N, int K, int m0, int n0, ALD a_ld, AXF a_xf, BLD b_ld, BXF b_xf, float* __restrict__ lds,
                                                       MID mid = MID()) {
    static_assert(WM * WN * WK == 4 || WM * WN * WK == 8, "4 or 8 waves per workgroup");
    typedef LtgRg<TM, TN, WM, WN, WK> G;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int nblk = (K + 31) >> 5;
    const int per = (nblk + WK - 1) / WK;
    const int Kc = K >= 4 ? ((K - 1) & ~3) : 0;   // start of the last (possibly partial) group of 4
    int am[TM], bn[TN];
    bool aok[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 16 + r;
        aok[tm] = m < M;
        am[tm] = min(m, M - 1);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bn[tn] = min(n0 + (wn * TN + tn) * 16 + r, N - 1);
    ltg_f32x4 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    auto pack8 = [] __device__(ltg_f32x4 lo, ltg_f32x4 hi) -> ltg_bf16x8 {
        typedef unsigned ltg_rg_u32x4 __attribute__((ext_vector_type(4)));
        ltg_rg_u32x4 p;     // (the values are bf16-representable: the upper halves ARE the bf16 bits)
        p[0] = (__float_as_uint(lo[0]) >> 16) | (__float_as_uint(lo[1]) & 0xFFFF0000u);
        p[1] = (__float_as_uint(lo[2]) >> 16) | (__float_as_uint(lo[3]) & 0xFFFF0000u);
        p[2] = (__float_as_uint(hi[0]) >> 16) | (__float_as_uint(hi[1]) & 0xFFFF0000u);
        p[3] = (__float_as_uint(hi[2]) >> 16) | (__float_as_uint(hi[3]) & 0xFFFF0000u);
        return __builtin_bit_cast(ltg_bf16x8, p);
    };
    auto pass = [&] __device__(const int base) {      // (first pass peeled: see ltg_rgemm_product)
        decltype(a_ld(0, 0, 0)) ra[NBLK][TM][2];
        decltype(b_ld(0, 0, 0)) rb[NBLK][TN][2];
        // phase 1: requests only
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kc = min(32 * (wk * per + base + i) + 8 * q + 4 * h, Kc);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) ra[i][tm][h] = a_ld(2 * i + h, am[tm], kc);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) rb[i][tn][h] = b_ld(2 * i + h, kc, bn[tn]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (base == 0) mid();
        // phase 2: transforms and MFMAs, block by block as the data lands
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            if (base + i >= per || 32 * (wk * per + base + i) >= K) continue;   // wave-uniform: nothing of this block is in range
            ltg_f32x4 av[TM][2], bv[TN][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 32 * (wk * per + base + i) + 8 * q + 4 * h;
                const bool ok = k < K;
                const int kc = min(k, Kc);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const ltg_f32x4 v = a_xf(ra[i][tm][h], 2 * i + h, am[tm], kc);
                    const bool o = ok && aok[tm];
                    av[tm][h] = ltg_f32x4{o ? v[0] : 0.f, o ? v[1] : 0.f, o ? v[2] : 0.f, o ? v[3] : 0.f};
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[tn][h] = b_xf(rb[i][tn][h], 2 * i + h, kc, bn[tn]);
            }
            ltg_bf16x8 ap[TM], bp[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) ap[tm] = pack8(av[tm][0], av[tm][1]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bp[tn] = pack8(bv[tn][0], bv[tn][1]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[tm], bp[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    if constexpr (PEEL) {
        pass(0);
        for (int base = NBLK; base < per; base += NBLK) pass(base);
    } else {
        for (int base = 0; base < per; base += NBLK) pass(base);
    }
    float* mine = lds + wk * (G::BM * G::LDC);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int x = 0; x < 4; ++x)
                mine[((wm * TM + tm) * 16 + 4 * q + x) * G::LDC + (wn * TN + tn) * 16 + r] = acc[tm][tn][x];
    __syncthreads();
}

// BFM: the product on the bf16 matrix pipe (ltg_rgemm_product_bf16; NBLK then counts 32-deep blocks)
// SPL: the fp32 product as bf16 cross terms of the split operands (ltg_rgemm_product; not with BFM)
template <int TM, int TN, int WM, int WN, int WK, int NBLK, bool BFM = false, bool PEEL = false, int SID = 0, int SPL = 0, class ALD, class AXF, class BLD, class BXF, class EF, class MID = LtgNoMid>
__device__ __forceinline__ void ltg_rgemm(int M, int N, int K, int m0, int n0, ALD a_ld, AXF a_xf, BLD b_ld, BXF b_xf, EF epi, float* __restrict__ lds,
                                          MID mid = MID()) {
    typedef LtgRg<TM, TN, WM, WN, WK> G;
    static_assert(!(BFM && SPL != 0), "operands that are bf16-rounded anyway need no split");
    if constexpr (BFM) ltg_rgemm_product_bf16<TM, TN, WM, WN, WK, NBLK, PEEL>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, lds, mid);
    else
    ltg_rgemm_product<TM, TN, WM, WN, WK, NBLK, PEEL, SID, SPL>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, lds, mid);
    const int tid = threadIdx.x;
    constexpr int NE = G::BM * G::BN;
    constexpr int NTH = 64 * WM * WN * WK;      // threads of the workgroup (256, or 512 with eight K slices)
    static_assert(NE % NTH == 0 || (NE < NTH && NE % 64 == 0), "tile must divide over the workgroup's threads, or be finished by its first waves");
    if (NE < NTH && tid >= NE) return;          // (wave-uniform: a 16 x 16 tile over eight K slices is finished by the first four waves)
#pragma unroll
    for (int e = 0; e < (NE + NTH - 1) / NTH; ++e) {
        const int id = tid + NTH * e;
        const int mm = id / G::BN, nn = id % G::BN;
        float v = lds[mm * G::LDC + nn];
#pragma unroll
        for (int s = 1; s < WK; ++s) v += lds[s * (G::BM * G::LDC) + mm * G::LDC + nn];
        const int m = m0 + mm, n = n0 + nn;
        epi(e, m, n, v, m < M && n < N);
    }
    LTG_STAMP_AT(SID, 5);
    LTG_STAMP_AT(SID, 6);
}

// The same block with a float4 epilogue: epi4(pre, m, n, value4, in_range) with n % 4 == 0; in_range = m < M && n < N (a
// ragged last group -- N % 4 != 0 -- is the caller's business).  pre = prefetch(m, n, in_range) is evaluated for the same
// (m, n) BEFORE the product: the epilogue's own operands (theta / m / v of an Adam update) are requested first, so the
// whole workgroup costs one memory round trip.
template <int TM, int TN, int WM, int WN, int WK, int NBLK, bool BFM = false, int SPL = 0, class ALD, class AXF, class BLD, class BXF, class PF, class EF>
__device__ __forceinline__ void ltg_rgemm_v4(int M, int N, int K, int m0, int n0, ALD a_ld, AXF a_xf, BLD b_ld, BXF b_xf, PF prefetch, EF epi4,
                                             float* __restrict__ lds) {
    typedef LtgRg<TM, TN, WM, WN, WK> G;
    const int tid = threadIdx.x;
    constexpr int NE4 = G::BM * G::BN / 4;
    constexpr int NP = (NE4 + 255) / 256;
    static_assert(NE4 % 256 == 0, "tile must divide over 256 threads in float4");
    decltype(prefetch(0, 0, false)) pre[NP];
    // (round 5: the epilogue's operands are requested in the product's mid hook -- BEHIND the first pass's operand requests, in front of anything
    // that consumes them -- and no longer in front of the product: the compiler drained them (s_waitcnt vmcnt(0)) before the first operand request
    // of some instantiations, a round trip of its own per tile.  In the queue behind the operands they cost nothing: the MFMAs wait with counts.)
    auto pre_mid = [&] __device__() {
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int id = tid + 256 * e;
            const int mm = id / (G::BN / 4), nn = (id % (G::BN / 4)) * 4;
            pre[e] = prefetch(m0 + mm, n0 + nn, m0 + mm < M && n0 + nn < N);
        }
    };
    if constexpr (BFM) ltg_rgemm_product_bf16<TM, TN, WM, WN, WK, NBLK>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, lds, pre_mid);
    else
    ltg_rgemm_product<TM, TN, WM, WN, WK, NBLK, false, 0, SPL>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, lds, pre_mid);
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        const int id = tid + 256 * e;
        const int mm = id / (G::BN / 4), nn = (id % (G::BN / 4)) * 4;
        ltg_f32x4 v = *reinterpret_cast<const ltg_f32x4*>(&lds[mm * G::LDC + nn]);
#pragma unroll
        for (int s = 1; s < WK; ++s) v += *reinterpret_cast<const ltg_f32x4*>(&lds[s * (G::BM * G::LDC) + mm * G::LDC + nn]);
        epi4(pre[e], m0 + mm, n0 + nn, v, m0 + mm < M && n0 + nn < N);
    }
}
