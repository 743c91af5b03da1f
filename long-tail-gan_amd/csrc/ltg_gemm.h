// Block-level MFMA GEMM building block for gfx950 (wave64).
//
// C[m][n] = sum_k A(m,k) * B(k,n), one BM x BN tile per 256-thread workgroup (4 waves arranged
// WM x WN), operands fetched through element functors (so gathers, transposes, on-the-fly
// derivatives and ones-augmentation for bias gradients are all "just a loader"), staged in LDS as
// k-contiguous rows, consumed by
//   MODE 1 (true) : v_mfma_f32_16x16x32_bf16      (operands rounded to bf16 RNE, fp32 accumulate)
//   MODE 0 (false): v_mfma_f32_16x16x4_f32        (exact fp32 fma chain)
//   MODE 2        : v_mfma_f32_16x16x32_fp8_fp8   (operands = OCP e4m3 of clamp(x * 2^SA | 2^SB, +-448), fp32 accumulate;
//                                                  the accumulator is scaled back by 2^-(SA+SB) before the epilogue)
// and handed to an epilogue functor epi(m, n, acc).
//
// Fragment maps (cdna_hip_programming.md section 3):
//   16x16x32 bf16: lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15], j=0..7  (fp8: same, one byte each)
//   16x16x4  f32 : lane l holds A[row l&15][k = l>>4],       B[k = l>>4][col l&15]
//   C/D (both)   : col = l&15, row = 4*(l>>4) + reg
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float ltg_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 ltg_bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short ltg_u16x8;

__device__ __forceinline__ unsigned short ltg_f2bf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
    return (unsigned short)(u >> 16);
}

// float -> OCP e4m3 (gfx950), round to nearest even; |x| is clamped to the format's largest finite value, 448, first
// (the ones-augmented rows of the bias gradients carry 1 * 2^8 = 256)
__device__ __forceinline__ unsigned char ltg_f2fp8(float x) {
    const float c = fminf(fmaxf(x, -448.f), 448.f);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.f, 0, false) & 0xFF);
}

template <int MODE, int BK_> struct LtgGemmCfg;
template <int BK_> struct LtgGemmCfg<1, BK_> {
    typedef unsigned short T;
    static constexpr int BK = BK_;
    static constexpr int LDK = BK_ + 8;  // rows stay 16-B aligned for the 16-byte fragment reads
    static_assert(BK_ % 32 == 0, "bf16 K-step is 32");
};
template <int BK_> struct LtgGemmCfg<0, BK_> {
    typedef float T;
    static constexpr int BK = BK_;
    static constexpr int LDK = BK_ + 1;  // odd word stride: conflict-free column walks
    static_assert(BK_ % 4 == 0, "fp32 K-step is 4");
};
template <int BK_> struct LtgGemmCfg<2, BK_> {
    typedef unsigned char T;
    static constexpr int BK = BK_;
    static constexpr int LDK = BK_ + 8;  // rows stay 8-B aligned for the 8-byte fragment reads
    static_assert(BK_ % 32 == 0, "fp8 K-step is 32");
};

// M, N: logical bounds of the tile grid (rows of A / columns of B); loaders are only called in range.
// [kbeg, kend): K range of this block (split-K).  A_MCONTIG / B_NCONTIG choose the thread->element
// map of the global loads so that consecutive threads walk the operand's contiguous dimension.
// VLOAD (bit 0: operand A, bit 1: operand B): that functor returns FOUR consecutive elements of the contiguous dimension
// (float4; the caller clamps the address and zeroes what lies outside the operand), 16 B per lane instead of 4:
//   a(m, k): A_MCONTIG ? (m..m+3, k) : (m, k..k+3)      b(k, n): B_NCONTIG ? (k, n..n+3) : (k..k+3, n)
template <int MODE, int BM, int BN, int BK_, int WM, int WN, bool A_MCONTIG, bool B_NCONTIG, bool VEC_EPI = false, int SA = 0, int SB = 0, int VLOAD = 0, class AF, class BF, class EF>
__device__ __forceinline__ void ltg_gemm_block(int M, int N, int m0, int n0, int kbeg, int kend, AF a, BF b, EF epi) {
    constexpr bool BF16 = MODE == 1, FP8 = MODE == 2;
    typedef LtgGemmCfg<MODE, BK_> Cfg;
    typedef typename Cfg::T T;
    constexpr int BK = Cfg::BK, LDK = Cfg::LDK;
    constexpr int NT = 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int EA = BM * BK / NT, EB = BN * BK / NT;
    static_assert(BM * BK % NT == 0 && BN * BK % NT == 0, "tile must divide over 256 threads");

    // one LDS block: operand tiles during the K loop, re-used as the fp32 output tile by VEC_EPI
    constexpr int OPER_BYTES = (BM + BN) * LDK * (int)sizeof(T);
    constexpr int CS_BYTES = VEC_EPI ? BM * (BN + 4) * 4 : 0;
    constexpr int SMEM_BYTES = OPER_BYTES > CS_BYTES ? OPER_BYTES : CS_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    T* const As = reinterpret_cast<T*>(smem);
    T* const Bs = As + BM * LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int lr = lane & 15, lq = lane >> 4;

    ltg_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr bool VA = (VLOAD & 1) != 0, VB = (VLOAD & 2) != 0;
    static_assert(!VLOAD || (BM % 4 == 0 && BN % 4 == 0 && BK % 4 == 0 && EA % 4 == 0 && EB % 4 == 0), "vector loads need multiples of 4");
    float ra[EA], rb[EB];
    // Loads are UNCONDITIONAL: indices are clamped into [0,M) x [kbeg,kend) (resp. [0,N)) and the value
    // is zeroed by a select afterwards, so the loaders never branch (a divergent branch per element
    // costs ~25 instructions and serialises dependent gathers).  Loaders may assume in-range indices.
    auto fetch = [&](int k0) {
        if constexpr (VA) {
#pragma unroll
            for (int j = 0; j < EA / 4; ++j) {
                const int v = tid + NT * j;
                const int mm = A_MCONTIG ? (v % (BM / 4)) * 4 : v / (BK / 4);
                const int kk = A_MCONTIG ? v / (BM / 4) : (v % (BK / 4)) * 4;
                const float4 x = a(m0 + mm, k0 + kk);
                ra[4 * j] = x.x; ra[4 * j + 1] = x.y; ra[4 * j + 2] = x.z; ra[4 * j + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EA; ++j) {
                const int e = tid + NT * j;
                const int mm = A_MCONTIG ? (e % BM) : (e / BK);
                const int kk = A_MCONTIG ? (e / BM) : (e % BK);
                const int gk = k0 + kk, gm = m0 + mm;
                const float v = a(min(gm, M - 1), min(gk, kend - 1));
                ra[j] = (gk < kend && gm < M) ? v : 0.f;
            }
        }
        if constexpr (VB) {
#pragma unroll
            for (int j = 0; j < EB / 4; ++j) {
                const int v = tid + NT * j;
                const int nn = B_NCONTIG ? (v % (BN / 4)) * 4 : v / (BK / 4);
                const int kk = B_NCONTIG ? v / (BN / 4) : (v % (BK / 4)) * 4;
                const float4 x = b(k0 + kk, n0 + nn);
                rb[4 * j] = x.x; rb[4 * j + 1] = x.y; rb[4 * j + 2] = x.z; rb[4 * j + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int e = tid + NT * j;
                const int nn = B_NCONTIG ? (e % BN) : (e / BK);
                const int kk = B_NCONTIG ? (e / BN) : (e % BK);
                const int gk = k0 + kk, gn = n0 + nn;
                const float v = b(min(gk, kend - 1), min(gn, N - 1));
                rb[j] = (gk < kend && gn < N) ? v : 0.f;
            }
        }
    };
    auto cvtA = [&](float x) -> T {
        if constexpr (BF16) return ltg_f2bf(x);
        else if constexpr (FP8) return ltg_f2fp8(x * (float)(1 << SA));
        else return x;
    };
    auto cvtB = [&](float x) -> T {
        if constexpr (BF16) return ltg_f2bf(x);
        else if constexpr (FP8) return ltg_f2fp8(x * (float)(1 << SB));
        else return x;
    };
    auto stash = [&]() {
        if constexpr (VA) {
#pragma unroll
            for (int j = 0; j < EA / 4; ++j) {
                const int v = tid + NT * j;
                const int mm = A_MCONTIG ? (v % (BM / 4)) * 4 : v / (BK / 4);
                const int kk = A_MCONTIG ? v / (BM / 4) : (v % (BK / 4)) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) As[(mm + (A_MCONTIG ? i : 0)) * LDK + kk + (A_MCONTIG ? 0 : i)] = cvtA(ra[4 * j + i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < EA; ++j) {
                const int e = tid + NT * j;
                const int mm = A_MCONTIG ? (e % BM) : (e / BK);
                const int kk = A_MCONTIG ? (e / BM) : (e % BK);
                As[mm * LDK + kk] = cvtA(ra[j]);
            }
        }
        if constexpr (VB) {
#pragma unroll
            for (int j = 0; j < EB / 4; ++j) {
                const int v = tid + NT * j;
                const int nn = B_NCONTIG ? (v % (BN / 4)) * 4 : v / (BK / 4);
                const int kk = B_NCONTIG ? v / (BN / 4) : (v % (BK / 4)) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[(nn + (B_NCONTIG ? i : 0)) * LDK + kk + (B_NCONTIG ? 0 : i)] = cvtB(rb[4 * j + i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int e = tid + NT * j;
                const int nn = B_NCONTIG ? (e % BN) : (e / BK);
                const int kk = B_NCONTIG ? (e / BN) : (e % BK);
                Bs[nn * LDK + kk] = cvtB(rb[j]);
            }
        }
    };

    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        stash();
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);  // next tile's loads fly under the MFMAs
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 32) {
                ltg_bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const ltg_u16x8 t = *reinterpret_cast<const ltg_u16x8*>(&As[(wm * WTM + i * 16 + lr) * LDK + ks + 8 * lq]);
                    af[i] = __builtin_bit_cast(ltg_bf16x8, t);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const ltg_u16x8 t = *reinterpret_cast<const ltg_u16x8*>(&Bs[(wn * WTN + j * 16 + lr) * LDK + ks + 8 * lq]);
                    bfr[j] = __builtin_bit_cast(ltg_bf16x8, t);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (FP8) {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 32) {
                long af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const long*>(&As[(wm * WTM + i * 16 + lr) * LDK + ks + 8 * lq]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const long*>(&Bs[(wn * WTN + j * 16 + lr) * LDK + ks + 8 * lq]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(wm * WTM + i * 16 + lr) * LDK + kk + lq];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = Bs[(wn * WTN + j * 16 + lr) * LDK + kk + lq];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    if constexpr (FP8) {
        constexpr float inv = 1.f / (float)(1 << (SA + SB));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= inv;
    }
    if constexpr (VEC_EPI) {
        // The accumulator tile takes a round trip through LDS so that the epilogue walks the output
        // row-major in float4 (16 B per lane, whole 128-B lines per row): epi(m, n, float4) with n % 4 == 0.
        constexpr int LDC = BN + 4;
        float* const Cs = reinterpret_cast<float*>(smem);
        __syncthreads();  // every wave is done reading the operand tiles
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(wm * WTM + i * 16 + lq * 4 + r) * LDC + wn * WTN + j * 16 + lr] = acc[i][j][r];
        __syncthreads();
        constexpr int N4 = BN / 4;
#pragma unroll
        for (int e = tid; e < BM * N4; e += NT) {
            const int mm = e / N4, c4 = e % N4;
            const int m = m0 + mm, n = n0 + c4 * 4;
            if (m < M && n < N) epi(m, n, *reinterpret_cast<const float4*>(&Cs[mm * LDC + c4 * 4]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + lq * 4 + r;
                    const int n = n0 + wn * WTN + j * 16 + lr;
                    if (m < M && n < N) epi(m, n, acc[i][j][r]);
                }
    }
}
