// Part of csrc/ltg_fast.h (included there, in this order, inside ltg_kernels.hip's anonymous namespace): latency-path kernels of small item slabs (I <= 4096): fk_dec1, fk_row_dlogits, fk_dh2, fk_g_tail.
// Split out of the 2 100-line header in round 6 -- the code is unchanged.
#pragma once

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
#ifndef LTG_TAIL_SPL
#define LTG_TAIL_SPL 6      // the fp32 tiles' products (dW_q0 of small slabs, dW_q1, dW_p0: 2 x 2 tiles per wave, K = the batch rows) as six bf16 cross terms of
                            // the split operands (ltg_rgemm.h; fp32-accurate): G phase 63.27 -> 62.93 ms per epoch of Askubuntu_Sample, same box
                            // (profiles/r6_ab_generator_split.txt); 0 = v_mfma_f32_16x16x4_f32
#endif
    if constexpr (RND) ltg_rgemm_v4<2, LTG_TAIL_BN / 16, 1, 1, 4, 1, true>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, prefetch, epi4, lds);
    else ltg_rgemm_v4<2, LTG_TAIL_BN / 16, 1, 1, 4, 2, false, LTG_TAIL_SPL>(M, N, K, m0, n0, a_ld, a_xf, b_ld, b_xf, prefetch, epi4, lds);
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
