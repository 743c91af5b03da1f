// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): generator step: row statistics, losses, dlogits and the generic backward kernels (train.py:145-164).
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------
// Generator step: losses (train.py:145-157) and backward (closed forms: SURVEY 8 row a10)
// ---------------------------------------------------------------------------------------------

// Per-row partial statistics over THIS rank's item slab (5 floats per row):
//   [0] m  = max_i logit            [1] s  = sum_i exp(logit - m)
//   [2] xl = sum_i x_bi * logit     [3] ps = sum_{(b,i) in S, i local} exp(logit - m)     [4] nx = sum_i x_bi
// Fake pairs carry GLOBAL item ids.  The shards' partials are all-gathered and combined in k_g_combine.
constexpr int RP = 5;
__global__ __launch_bounds__(NT) void k_row_partial(int I, int item_lo, const int32_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                    const float* __restrict__ logits, int nf, const int32_t* __restrict__ f_row,
                                                    const int32_t* __restrict__ f_gen, const int32_t* __restrict__ f_pop,
                                                    float* __restrict__ rowpart) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const float* row = logits + (size_t)b * I;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < I; i += NT) mx = fmaxf(mx, row[i]);
    mx = block_max(mx, red);
    float s = 0.f, xl = 0.f, nx = 0.f, ps = 0.f;
    for (int i = threadIdx.x; i < I; i += NT) s += expf(row[i] - mx);
    for (int e = indptr[b] + threadIdx.x; e < indptr[b + 1]; e += NT) {
        const float v = values ? values[e] : 1.f;
        xl += v * row[indices[e]];
        nx += v;
    }
    for (int q = threadIdx.x; q < nf; q += NT) {
        const int it = f_gen[q] - item_lo;
        if (f_row[q] == b && f_gen[q] >= 0 && f_pop[q] >= 0 && it >= 0 && it < I) ps += expf(row[it] - mx);
    }
    s = block_sum(s, red);
    xl = block_sum(xl, red);
    nx = block_sum(nx, red);
    ps = block_sum(ps, red);
    if (threadIdx.x == 0) {
        float* o = rowpart + (size_t)b * RP;
        o[0] = mx;
        o[1] = s;
        o[2] = xl;
        o[3] = ps;
        o[4] = nx;
    }
}

// Large item slabs: the same statistics per (4096-item segment, row) in one pass over the logits (the segment
// lives in registers between the max and the exp-sum), merged per row by k_row_partial_merge.
constexpr int RS_SEG = 4096;
// scratch of the row statistics: 5 floats per (4096-item segment, row) for k_row_partial_seg, or 2 floats per (workgroup of
// k_dec1_fwd_stream<true>, row) -- at most 256 workgroups
inline size_t segpart_floats(int I, int rows) {
    const size_t a = ((size_t)I + RS_SEG - 1) / RS_SEG * rows * 5, b = (size_t)256 * rows * 2;
    return a > b ? a : b;
}
__global__ __launch_bounds__(NT) void k_row_partial_seg(int I, int item_lo, const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                        const float* __restrict__ logits, int nf, const int32_t* __restrict__ f_row,
                                                        const int32_t* __restrict__ f_gen, const int32_t* __restrict__ f_pop,
                                                        float* __restrict__ segpart) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.y, sg = blockIdx.x, B = gridDim.y;
    const int i0 = sg * RS_SEG, i1 = min(I, i0 + RS_SEG);
    const float* row = logits + (size_t)b * I;
    float v[RS_SEG / NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < RS_SEG / NT; ++j) {
        const int i = i0 + threadIdx.x + NT * j;
        v[j] = i < i1 ? row[i] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = block_max(mx, red);
    float s = 0.f, xl = 0.f, nx = 0.f, ps = 0.f;
#pragma unroll
    for (int j = 0; j < RS_SEG / NT; ++j) s += expf(v[j] - mx);  // exp(-inf) = 0 for the tail
    for (int e = indptr[b] + threadIdx.x; e < indptr[b + 1]; e += NT) {
        const int it = indices[e];
        if (it >= i0 && it < i1) {
            const float x = values ? values[e] : 1.f;
            xl += x * row[it];
            nx += x;
        }
    }
    for (int q = threadIdx.x; q < nf; q += NT) {
        const int it = f_gen[q] - item_lo;
        if (f_row[q] == b && f_gen[q] >= 0 && f_pop[q] >= 0 && it >= i0 && it < i1) ps += expf(row[it] - mx);
    }
    s = block_sum(s, red);
    xl = block_sum(xl, red);
    nx = block_sum(nx, red);
    ps = block_sum(ps, red);
    if (threadIdx.x == 0) {
        float* o = segpart + ((size_t)sg * B + b) * RP;
        o[0] = mx;
        o[1] = s;
        o[2] = xl;
        o[3] = ps;
        o[4] = nx;
    }
}

// merge the segment partials of a row into one partial (same 5-float format); optionally also the row's lse
__global__ __launch_bounds__(64) void k_row_partial_merge(int B, int nseg, const float* __restrict__ segpart,
                                                          float* __restrict__ rowpart, float* __restrict__ lse) {
    // one wave per row, lanes over the segments (49 at 200 000 items): every partial is requested at once
    const int b = blockIdx.x, lane = threadIdx.x;
    float M = -INFINITY;
    for (int g = lane; g < nseg; g += 64) M = fmaxf(M, segpart[((size_t)g * B + b) * RP]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
    float s = 0.f, xl = 0.f, ps = 0.f, nx = 0.f;
    for (int g = lane; g < nseg; g += 64) {
        const float* q = segpart + ((size_t)g * B + b) * RP;
        const float sc = expf(q[0] - M);
        s += q[1] * sc;
        xl += q[2];
        ps += q[3] * sc;
        nx += q[4];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        xl += __shfl_xor(xl, o);
        ps += __shfl_xor(ps, o);
        nx += __shfl_xor(nx, o);
    }
    if (lane == 0) {
        if (rowpart) {
            float* o = rowpart + (size_t)b * RP;
            o[0] = M;
            o[1] = s;
            o[2] = xl;
            o[3] = ps;
            o[4] = nx;
        }
        if (lse) lse[b] = M + logf(s);
    }
}

// Row partial (same 5 floats) from the statistics k_dec1_fwd_stream<true> left: fold the G workgroups' (max, sum exp) pairs
// of the row, then the sparse terms -- sum x logit and sum x over the row's entries, sum exp(logit - max) over its fake pairs
__global__ __launch_bounds__(NT) void k_row_stats_merge(int B, int G, int I, int item_lo, const float* __restrict__ stat, const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                        const float* __restrict__ logits, int nf, const int32_t* __restrict__ f_row,
                                                        const int32_t* __restrict__ f_gen, const int32_t* __restrict__ f_pop,
                                                        float* __restrict__ rowpart, float* __restrict__ lse) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (size_t)b * I;
    // Round 5, second pass: TWO levels of requests for the common case (G <= NT, a row of at most NT entries, nf <= 4 NT fake pairs) -- level 1:
    // the thread's (max, sum exp) pair, its first sparse entry and its first batch of fake-pair triples; level 2: the logits those point at.
    // As written before -- max loop, barrier, the pairs AGAIN, indptr -> indices -> logit, triples -> logit -- the row cost seven dependent trips.
    // Every sum below adds the same terms in the same order as before.
    constexpr int FU = 4;
    const int e0 = indptr[b], e1 = indptr[b + 1];
    float st0 = -INFINITY, st1 = 0.f;
    if (tid < G) {
        const float* q = stat + ((size_t)tid * B + b) * 2;
        st0 = q[0];
        st1 = q[1];
    }
    int tg0[FU], tr0[FU], tp0[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) tg0[u] = tr0[u] = tp0[u] = -1;
    if (nf > 0) {      // (ONE uniform branch around the twelve requests: a select per element made a basic block -- and a wait -- of each)
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = min(tid + u * NT, nf - 1);
            tg0[u] = f_gen[q];
            tr0[u] = f_row[q];
            tp0[u] = f_pop[q];
        }
    }
    int idx0 = -1;
    float x0 = 1.f;
    if (e0 + tid < e1) {
        idx0 = indices[e0 + tid];
        if (values) x0 = values[e0 + tid];
    }
    // level 2
    const float rl0 = row[max(idx0, 0)];
    float lg0[FU];
    bool ok0[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
        const int it = tg0[u] - item_lo;
        ok0[u] = tid + u * NT < nf && tr0[u] == b && tg0[u] >= 0 && tp0[u] >= 0 && it >= 0 && it < I;
        lg0[u] = row[ok0[u] ? it : 0];
    }
    float mx = fmaxf(-INFINITY, st0);
    for (int g = tid + NT; g < G; g += NT) mx = fmaxf(mx, stat[((size_t)g * B + b) * 2]);
    mx = block_max(mx, red);
    float s = 0.f, xl = 0.f, nx = 0.f, ps = 0.f;
    if (tid < G) s += st1 * expf(st0 - mx);
    for (int g = tid + NT; g < G; g += NT) {
        const float* q = stat + ((size_t)g * B + b) * 2;
        s += q[1] * expf(q[0] - mx);
    }
    if (idx0 >= 0) {
        xl += x0 * rl0;
        nx += x0;
    }
    for (int e = e0 + tid + NT; e < e1; e += NT) {
        const float x = values ? values[e] : 1.f;
        xl += x * row[indices[e]];
        nx += x;
    }
#pragma unroll
    for (int u = 0; u < FU; ++u)
        if (ok0[u]) ps += expf(lg0[u] - mx);
    // (the thread's further fake pairs in batches of four -- the triples requested together, then the logits of the pairs that count,
    // added in the loop's order)
    for (int q0 = tid + FU * NT; q0 < nf; q0 += FU * NT) {
        int tg[FU], tr[FU], tp[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = min(q0 + u * NT, nf - 1);
            tg[u] = f_gen[q];
            tr[u] = f_row[q];
            tp[u] = f_pop[q];
        }
        float lg[FU];
        bool ok[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int it = tg[u] - item_lo;
            ok[u] = q0 + u * NT < nf && tr[u] == b && tg[u] >= 0 && tp[u] >= 0 && it >= 0 && it < I;
            lg[u] = row[ok[u] ? it : 0];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (ok[u]) ps += expf(lg[u] - mx);
    }
    s = block_sum(s, red);
    xl = block_sum(xl, red);
    nx = block_sum(nx, red);
    ps = block_sum(ps, red);
    if (threadIdx.x == 0) {
        if (rowpart) {
            float* o = rowpart + (size_t)b * RP;
            o[0] = mx;
            o[1] = s;
            o[2] = xl;
            o[3] = ps;
            o[4] = nx;
        }
        if (lse) lse[b] = mx + logf(s);
    }
}

// The R shards' partials of row rb folded into (lse, n_b, P_b, sum x logit): max over the ranks, then the sums in ascending rank order
// (k_g_combine's arithmetic and order).  Round 5: for R <= 8 the R x 5 floats are requested AT ONCE (clamped, masked) -- as three plain loops
// over a runtime rank count every partial was a round trip of its own, 3 R of them in front of the first logit k_dlogits_combine reads
// (24 at eight ranks).
__device__ __forceinline__ void ltg_rank_terms(const float* __restrict__ rowpart_all, int R, int B, int rb, float& l, float& nx, float& pb, float& xl) {
    constexpr int RU = 8;
    if (R == 1) {      // one rank: the five floats, the same operations in the same order as the general form
        const float* q = rowpart_all + (size_t)rb * RP;
        const float a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
        const float M = fmaxf(-INFINITY, a0);
        const float se = 0.f + a1 * expf(a0 - M);
        xl = 0.f + a2;
        nx = 0.f + a4;
        l = M + logf(se);
        pb = 0.f + a3 * expf(a0 - l);
        return;
    }
    if (R <= RU) {
        float q0[RU], q1[RU], q2[RU], q3[RU], q4[RU];
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            const float* q = rowpart_all + ((size_t)min(r, R - 1) * B + rb) * RP;
            q0[r] = q[0]; q1[r] = q[1]; q2[r] = q[2]; q3[r] = q[3]; q4[r] = q[4];
        }
        float M = -INFINITY;
#pragma unroll
        for (int r = 0; r < RU; ++r)
            if (r < R) M = fmaxf(M, q0[r]);
        float se = 0.f;
        xl = 0.f;
        nx = 0.f;
#pragma unroll
        for (int r = 0; r < RU; ++r)
            if (r < R) {
                se += q1[r] * expf(q0[r] - M);
                xl += q2[r];
                nx += q4[r];
            }
        l = M + logf(se);
        pb = 0.f;
#pragma unroll
        for (int r = 0; r < RU; ++r)
            if (r < R) pb += q3[r] * expf(q0[r] - l);
        return;
    }
    float M = -INFINITY;
    for (int r = 0; r < R; ++r) M = fmaxf(M, rowpart_all[((size_t)r * B + rb) * RP]);
    float se = 0.f;
    xl = 0.f;
    nx = 0.f;
    for (int r = 0; r < R; ++r) {
        const float* q = rowpart_all + ((size_t)r * B + rb) * RP;
        se += q[1] * expf(q[0] - M);
        xl += q[2];
        nx += q[4];
    }
    l = M + logf(se);
    pb = 0.f;
    for (int r = 0; r < R; ++r) {
        const float* q = rowpart_all + ((size_t)r * B + rb) * RP;
        pb += q[3] * expf(q[0] - l);
    }
}

// Combine the R shards' row partials: lse, n_b, P_b per row, then the step scalars (train.py:145-157):
// out[0]=g_loss out[1]=vae_loss out[2]=gan_loss out[3]=sum_S p out[4]=sum_j y_j out[5]=c
__global__ __launch_bounds__(NT) void k_g_combine(int B, int R, const float* __restrict__ rowpart_all, int nf,
                                                  const float* __restrict__ kl_rows, const float* __restrict__ y,
                                                  const int32_t* __restrict__ cnt, float anneal, float lam, float* __restrict__ lse,
                                                  float* __restrict__ nb, float* __restrict__ Pb, float* __restrict__ out,
                                                  float* __restrict__ out2) {
    __shared__ float red[NT / 64];
    float a = 0.f, k = 0.f, p = 0.f, sy = 0.f;
    for (int b = threadIdx.x; b < B; b += NT) {
        float l, nx, pb, xl;
        ltg_rank_terms(rowpart_all, R, B, b, l, nx, pb, xl);
        lse[b] = l;
        nb[b] = nx;
        Pb[b] = pb;
        a += -xl + nx * l;  // neg_ll_row = -sum x (logit - lse)
        if (kl_rows) k += kl_rows[b];
        p += pb;
    }
    if (y)
        for (int i = threadIdx.x; i < nf; i += NT) sy += y[i];
    a = block_sum(a, red);
    k = block_sum(k, red);
    p = block_sum(p, red);
    sy = block_sum(sy, red);
    if (threadIdx.x == 0 && out) {
        const float negll = a / (float)B, KL = k / (float)B;
        const float c = (cnt && cnt[0] > 0) ? lam / (float)cnt[0] * sy : 0.f;
        const float vae = negll + anneal * KL;
        const float gan = -c * p;
        const float r[6] = {vae + gan, vae, gan, p, sy, c};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            out[i] = r[i];
            if (out2) out2[i] = r[i];  // the caller's loss buffer (no separate device-to-device copy)
        }
    }
}

// candidate logits of this rank's slab (0 elsewhere): summed over ranks they give every rank the
// logits of all candidates (the sampler needs nothing else of the [B, I] matrix)
__global__ __launch_bounds__(NT) void k_gather_cand(int I, int item_lo, const int32_t* __restrict__ cand_ptr,
                                                    const int32_t* __restrict__ cand_idx, const float* __restrict__ logits,
                                                    float* __restrict__ out) {
    const int b = blockIdx.x;
    for (int j = cand_ptr[b] + threadIdx.x; j < cand_ptr[b + 1]; j += NT) {
        const int it = cand_idx[j] - item_lo;
        out[j] = (it >= 0 && it < I) ? logits[(size_t)b * I + it] : 0.f;
    }
}

// dlogits[b][i] = p*(n_b/B + c*P_b) - x_bi/B - c*p*[(b,i) in S]; grid (segments, rows).
constexpr int DL_SEG = 2048;
__global__ __launch_bounds__(NT) void k_dlogits(int B, int I, const int32_t* __restrict__ indptr,
                                                const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                const float* __restrict__ logits, const float* __restrict__ lse,
                                                const float* __restrict__ nb, const float* __restrict__ Pb,
                                                const float* __restrict__ scal, int nf, const int32_t* __restrict__ f_row,
                                                const int32_t* __restrict__ f_gen, const int32_t* __restrict__ f_pop,
                                                float* __restrict__ dlog, int item_lo) {
    __shared__ float s_x[DL_SEG];
    __shared__ uint8_t s_s[DL_SEG];
    const int b = blockIdx.y, i0 = blockIdx.x * DL_SEG;
    const int i1 = min(I, i0 + DL_SEG);
    for (int j = threadIdx.x; j < DL_SEG; j += NT) {
        s_x[j] = 0.f;
        s_s[j] = 0;
    }
    __syncthreads();
    for (int e = indptr[b] + threadIdx.x; e < indptr[b + 1]; e += NT) {
        const int it = indices[e];
        if (it >= i0 && it < i1) s_x[it - i0] = values ? values[e] : 1.f;
    }
    for (int s = threadIdx.x; s < nf; s += NT) {
        const int it = f_gen[s] - item_lo;  // fake pairs carry global item ids
        if (f_row[s] == b && f_gen[s] >= 0 && it >= i0 && it < i1 && f_pop[s] >= 0) s_s[it - i0] = 1;
    }
    __syncthreads();
    const float invB = 1.f / (float)B, c = scal[5], l = lse[b];
    const float alpha = nb[b] * invB + c * Pb[b];
    const size_t base = (size_t)b * I;
    for (int i = i0 + threadIdx.x; i < i1; i += NT) {
        const float p = expf(logits[base + i] - l);
        dlog[base + i] = p * alpha - s_x[i - i0] * invB - (s_s[i - i0] ? c * p : 0.f);
    }
}

// The same with the combine of the R shards' row partials folded in (no k_g_combine launch in front): every workgroup merges
// the R x 5 partials of ITS row (uniform addresses: scalar loads) and adds up sum_j y_j itself; the segment-0 workgroups also
// publish lse, and workgroup (0, 0) the step's scalars (train.py:154-157).  Same arithmetic and order as k_g_combine.
// D16: dlog is stored as bf16 (the streaming consumers feed it to the bf16 MFMA anyway: same operand bits, half the bytes)
template <bool D16>
__global__ __launch_bounds__(NT) void k_dlogits_combine(int B, int I, int R, const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                        const float* __restrict__ logits, const float* __restrict__ rowpart_all,
                                                        const float* __restrict__ kl_rows, const float* __restrict__ y,
                                                        const int32_t* __restrict__ cnt, float anneal, float lam, int nf,
                                                        const int32_t* __restrict__ f_row, const int32_t* __restrict__ f_gen,
                                                        const int32_t* __restrict__ f_pop, float* __restrict__ dlog, float* __restrict__ lse,
                                                        float* __restrict__ out, float* __restrict__ out2, int item_lo) {
    __shared__ float s_x[DL_SEG];
    __shared__ uint8_t s_s[DL_SEG];
    __shared__ float red[NT / 64];
    const int b = blockIdx.y, i0 = blockIdx.x * DL_SEG, tid = threadIdx.x;
    const int i1 = min(I, i0 + DL_SEG);
    const size_t base = (size_t)b * I;
    // Round 5, second pass: every request that depends on nothing FIRST -- the segment's logits themselves (they were the LAST thing the kernel
    // asked for, one dependent trip per 512 items behind five others), the first batch of y's and fake-pair triples, the thread's first sparse
    // entry, the ranks' row partials and cnt[0].  Sums and stores are the same terms in the same order as before.
    constexpr int FU = 4;
    constexpr int NL = D16 ? DL_SEG / (2 * NT) : DL_SEG / NT;
    float2 lg2[D16 ? NL : 1];
    float lg1[D16 ? 1 : NL];
    if constexpr (D16) {
#pragma unroll
        for (int j = 0; j < NL; ++j) lg2[j] = *reinterpret_cast<const float2*>(logits + base + min(i0 + 2 * tid + 2 * NT * j, I - 2));   // (I % 8 == 0)
    } else {
#pragma unroll
        for (int j = 0; j < NL; ++j) lg1[j] = logits[base + min(i0 + tid + NT * j, I - 1)];
    }
    float ty0[FU];
    int tg0[FU], tr0[FU], tp0[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
        ty0[u] = 0.f;
        tg0[u] = tr0[u] = tp0[u] = -1;
    }
    if (nf > 0) {      // (ONE uniform branch around the requests: a select per element made a basic block -- and a wait -- of each)
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = min(tid + u * NT, nf - 1);
            tg0[u] = f_gen[q];
            tr0[u] = f_row[q];
            tp0[u] = f_pop[q];
        }
        if (y) {
#pragma unroll
            for (int u = 0; u < FU; ++u) ty0[u] = y[min(tid + u * NT, nf - 1)];
        }
    }
    const int e0 = indptr[b], e1 = indptr[b + 1];
    int it0 = -1;
    float x0 = 1.f;
    if (e0 + tid < e1) {
        it0 = indices[e0 + tid];
        if (values) x0 = values[e0 + tid];
    }
    auto row_terms = [=] __device__(int rb, float& l, float& nx, float& pb, float& xl) { ltg_rank_terms(rowpart_all, R, B, rb, l, nx, pb, xl); };
    float l, nx, pb, xl;
    row_terms(b, l, nx, pb, xl);
    const int cntv = cnt ? cnt[0] : 0;
    for (int j = tid; j < DL_SEG; j += NT) {
        s_x[j] = 0.f;
        s_s[j] = 0;
    }
    float sy = 0.f;
    if (y) {
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (tid + u * NT < nf) sy += ty0[u];
        for (int q0 = tid + FU * NT; q0 < nf; q0 += FU * NT) {
            float ty[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) ty[u] = y[min(q0 + u * NT, nf - 1)];
#pragma unroll
            for (int u = 0; u < FU; ++u)
                if (q0 + u * NT < nf) sy += ty[u];
        }
    }
    __syncthreads();
    if (it0 >= i0 && it0 < i1) s_x[it0 - i0] = x0;
    for (int e = e0 + tid + NT; e < e1; e += NT) {
        const int it = indices[e];
        if (it >= i0 && it < i1) s_x[it - i0] = values ? values[e] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
        const int it = tg0[u] - item_lo;  // fake pairs carry global item ids
        if (tid + u * NT < nf && tr0[u] == b && tg0[u] >= 0 && it >= i0 && it < i1 && tp0[u] >= 0) s_s[it - i0] = 1;
    }
    for (int q0 = tid + FU * NT; q0 < nf; q0 += FU * NT) {
        int tg[FU], tr[FU], tp[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = min(q0 + u * NT, nf - 1);
            tg[u] = f_gen[q];
            tr[u] = f_row[q];
            tp[u] = f_pop[q];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int it = tg[u] - item_lo;
            if (q0 + u * NT < nf && tr[u] == b && tg[u] >= 0 && it >= i0 && it < i1 && tp[u] >= 0) s_s[it - i0] = 1;
        }
    }
    sy = block_sum(sy, red);   // (its barriers also publish s_x / s_s)
    const float invB = 1.f / (float)B;
    const float c = cntv > 0 ? lam / (float)cntv * sy : 0.f;
    const float alpha = nx * invB + c * pb;
    if constexpr (D16) {   // I % 8 == 0 (stream_ok): pairs of items, one 8-B load and one 4-B store per lane
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + 2 * tid + 2 * NT * j;
            if (i < i1) {
                const float2 lg = lg2[j];
                const float p0 = expf(lg.x - l), p1 = expf(lg.y - l);
                const float d0 = p0 * alpha - s_x[i - i0] * invB - (s_s[i - i0] ? c * p0 : 0.f);
                const float d1 = p1 * alpha - s_x[i + 1 - i0] * invB - (s_s[i + 1 - i0] ? c * p1 : 0.f);
                reinterpret_cast<unsigned*>(dlog)[(base + i) >> 1] = (unsigned)ltg_f2bf(d0) | ((unsigned)ltg_f2bf(d1) << 16);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + tid + NT * j;
            if (i < i1) {
                const float p = expf(lg1[j] - l);
                dlog[base + i] = p * alpha - s_x[i - i0] * invB - (s_s[i - i0] ? c * p : 0.f);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) lse[b] = l;
    if (blockIdx.x == 0 && blockIdx.y == 0) {   // the step's scalars: out[0]=g_loss out[1]=vae_loss out[2]=gan_loss out[3]=sum_S p out[4]=sum_j y_j out[5]=c
        float a = 0.f, k = 0.f, pp = 0.f;
        for (int rb = threadIdx.x; rb < B; rb += NT) {
            float l2, nx2, pb2, xl2;
            row_terms(rb, l2, nx2, pb2, xl2);
            a += -xl2 + nx2 * l2;
            if (kl_rows) k += kl_rows[rb];
            pp += pb2;
        }
        a = block_sum(a, red);
        k = block_sum(k, red);
        pp = block_sum(pp, red);
        if (threadIdx.x == 0) {
            const float negll = a / (float)B, KL = k / (float)B;
            const float vae = negll + anneal * KL, gan = -c * pp;
            const float r6[6] = {vae + gan, vae, gan, pp, sy, c};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (out) out[i] = r6[i];
                if (out2) out2[i] = r6[i];
            }
        }
    }
}

// dh2 partials: part[z][b][h] = sum_{i in split z} dlog[b][i] * W_p1t[i][h]   (split-K over items)
template <bool BF16, bool BIG, bool V = false>
__global__ __launch_bounds__(NT) void k_dh2_partial(int B, int I, int H, int kchunk, const float* __restrict__ dlog,
                                                    const float* __restrict__ Wp1t, float* __restrict__ part) {
    constexpr int BM = BIG ? 128 : 32, BN = BIG ? 64 : 32;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * kchunk, kend = min(I, kbeg + kchunk);
    float* out = part + (size_t)blockIdx.z * B * H;
    auto a = [=] __device__(int m, int k) -> float { return dlog[(size_t)m * I + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Wp1t[(size_t)k * H + n]; };
    auto epi = [=] __device__(int m, int n, float acc) { out[(size_t)m * H + n] = acc; };
    if constexpr (V) {   // 16-B loaders (I % 4 == 0, H % 4 == 0; the K chunks are multiples of 32)
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(dlog + (size_t)min(m, B - 1) * I, k, kend, m < B); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Wp1t + (size_t)min(k, kend - 1) * H, n, H, k < kend); };
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, true, false, 0, 0, 3>(B, H, m0, n0, kbeg, kend, a4, b4, epi);
    } else {
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, true>(B, H, m0, n0, kbeg, kend, a, b, epi);
    }
}

// da2 = (sum_z part) * (1 - h2^2)
__global__ __launch_bounds__(NT) void k_da2(int n, int nsplit, const float* __restrict__ part, const float* __restrict__ h2,
                                            float* __restrict__ da2, LtgGate started = LTG_NO_GATE) {
    // started: opened by the first workgroup as soon as this kernel runs -- whatever preceded it on its stream (the dh2 product) is
    // complete, which is what the forked weight update waits for
    if (blockIdx.x == 0 && threadIdx.x == 0) ltg_gate_set(started);
    // one output per thread (B H = 60 000 outputs -> 235 workgroups instead of 59 with float4), 16 slabs in flight; the slabs
    // are added in ascending order whatever the unroll: bitwise the same sum as a serial walk
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        // (round 5: 32 slabs in flight and the remainder as ONE clamped, masked batch -- with 16 and a serial remainder the 98 slabs of a
        // 25 024-item step were eight dependent round trips in a 6-us launch on the caller's stream)
        float s = 0.f;
        constexpr int DU = 32;
        const float t = h2 ? h2[i] : 0.f;      // (requested with the first slabs, not behind them)
        for (int z = 0; z < nsplit; z += DU) {
            float x[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) x[u] = part[(size_t)min(z + u, nsplit - 1) * n + i];
#pragma unroll
            for (int u = 0; u < DU; ++u)
                if (z + u < nsplit) s += x[u];
        }
        da2[i] = s * __builtin_fmaf(-t, t, 1.f);   // (rounding pinned: fk_dz_dh2's operand loader computes the same expression)
    }
}

// dW_p1t[i][h] = sum_b dlog[b][i] h2[b][h]; column H = ones -> db_p1[i]; fused Adam on both.
template <bool BF16, int VAR, bool V = false, bool D16 = false>
__global__ __launch_bounds__(NT) void k_dec1_bwd_adam(int B, int I, int H, const float* __restrict__ dlog,
                                                      const float* __restrict__ h2, ltg_gen_state st, AdamC ad, int i_begin,
                                                      const unsigned* __restrict__ poison = nullptr) {
    if (ltg_poisoned(poison)) return;   // (the ragged tail of the one-call step's forked weight update)
    // VAR 0: 32x32 tiles, scalar Adam epilogue; 1: 64x128, 2: 64x64, 3: 32x128 tiles with the float4 epilogue
    constexpr bool BIG = VAR != 0;
    constexpr int BM = VAR == 0 ? 32 : (VAR == 3 ? 32 : 64), BN = VAR == 0 ? 32 : (VAR == 2 ? 64 : 128);
    const int m0 = i_begin + blockIdx.y * BM, n0 = blockIdx.x * BN;   // i_begin: first item row of this launch
    float *W = st.p[3], *mW = st.m[3], *vW = st.v[3], *bb = st.p[7], *mb = st.m[7], *vb = st.v[7];
    unsigned short* Wb = st.wp1t_bf16;  // optional bf16 shadow [I][ST_KP], kept in step with the master weights
    auto a = [=] __device__(int m, int k) -> float {
        if constexpr (D16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(dlog)[(size_t)k * I + m] << 16);   // dlog stored as bf16
        else return dlog[(size_t)k * I + m];
    };
    auto b = [=] __device__(int k, int n) -> float {
        const float v = h2[(size_t)k * H + min(n, H - 1)];
        return n < H ? v : 1.f;
    };
    if constexpr (BIG) {
        // Adam epilogue in float4 over whole 512-B row segments of W_p1t / m / v (H % 4 == 0)
        auto epi = [=] __device__(int m, int n, float4 g) {
            if (n < H) {
                const size_t o = ((size_t)m * H + n) >> 2;
                float4 p = reinterpret_cast<float4*>(W)[o], mm = reinterpret_cast<float4*>(mW)[o], vv = reinterpret_cast<float4*>(vW)[o];
#define LTG_ADAM4(f) adam1(p.f, mm.f, vv.f, g.f, ad.lr_t, ad);
                LTG_ADAM4(x) LTG_ADAM4(y) LTG_ADAM4(z) LTG_ADAM4(w)
#undef LTG_ADAM4
                reinterpret_cast<float4*>(W)[o] = p;
                reinterpret_cast<float4*>(mW)[o] = mm;
                reinterpret_cast<float4*>(vW)[o] = vv;
                if (Wb) *reinterpret_cast<uint2*>(Wb + (size_t)m * ST_KP + n) = ltg_pack4(p);
            } else {
                adam_update(bb, mb, vb, m, g.x, ad);  // n == H: the ones column = bias gradient
            }
        };
        ltg_gemm_block<BF16, BM, BN, 128, 2, 2, true, true, true>(I, H + 1, m0, n0, 0, B, a, b, epi);
    } else {
        auto epi = [=] __device__(int m, int n, float g) {
            if (n < H) {
                adam_update(W, mW, vW, (size_t)m * H + n, g, ad);
                if (Wb) Wb[(size_t)m * ST_KP + n] = ltg_f2bf(W[(size_t)m * H + n]);
            } else adam_update(bb, mb, vb, m, g, ad);
        };
        if constexpr (V) {   // 16-B loaders (I % 4 == 0, H % 4 == 0: the ones column n == H opens its own group)
            auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(dlog + (size_t)min(k, B - 1) * I, m, I, k < B); };
            auto b4 = [=] __device__(int k, int n) -> float4 {
                float4 v = ltg_ld4(h2 + (size_t)min(k, B - 1) * H, n, H, k < B);
                if (n == H && k < B) v.x = 1.f;
                return v;
            };
            ltg_gemm_block<BF16, BM, BN, 128, 2, 2, true, true, false, 0, 0, 3>(I, H + 1, m0, n0, 0, B, a4, b4, epi);
        } else {
            ltg_gemm_block<BF16, BM, BN, 128, 2, 2, true, true>(I, H + 1, m0, n0, 0, B, a, b, epi);
        }
    }
}

// dz = da2 . W_p0^T, then d mu / d logvar (KL + reparameterisation terms)
template <bool V, int BKV = 128>
__global__ __launch_bounds__(NT) void k_dz(int B, int Z, int H, const float* __restrict__ da2, const float* __restrict__ Wp0,
                                           const float* __restrict__ mulv, const float* __restrict__ eps_in, float is_training,
                                           float anneal, uint64_t seed, uint64_t step, float* __restrict__ dmlv) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float invB = 1.f / (float)B;
    auto a = [=] __device__(int m, int k) -> float { return da2[(size_t)m * H + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Wp0[(size_t)n * H + k]; };
    auto epi = [=] __device__(int m, int n, float dz) {
        const float mu = mulv[(size_t)m * 2 * Z + n], lv = mulv[(size_t)m * 2 * Z + Z + n];
        float e = 0.f;
        if (is_training != 0.f)
            e = eps_in ? eps_in[(size_t)m * Z + n] : ltg_rng_normal(seed, LTG_STREAM_VAE_EPS, step, (uint64_t)m * Z + n);
        dmlv[(size_t)m * 2 * Z + n] = dz + anneal * mu * invB;
        dmlv[(size_t)m * 2 * Z + Z + n] = dz * is_training * e * expf(0.5f * lv) * 0.5f + anneal * 0.5f * (expf(lv) - 1.f) * invB;
    };
    if constexpr (V) {
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(da2 + (size_t)min(m, B - 1) * H, k, H, m < B); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Wp0 + (size_t)min(n, Z - 1) * H, k, H, n < Z); };
        ltg_gemm_block<false, 32, 32, BKV, 2, 2, false, false, false, 0, 0, 3>(B, Z, m0, n0, 0, H, a4, b4, epi);
    } else {
        ltg_gemm_block<false, 32, 32, 128, 2, 2, false, false>(B, Z, m0, n0, 0, H, a, b, epi);
    }
}

// generic "weight gradient + Adam": G[m][n] = sum_k L(k,m) * R(k,n) with ones-augmented row m == Min
// (bias gradient).  L: [K][Min] activations, R: [K][N] upstream gradient.
template <bool V>
__global__ __launch_bounds__(NT) void k_wgrad_adam(int K, int Min, int N, const float* __restrict__ L,
                                                   const float* __restrict__ R, float* __restrict__ W, float* __restrict__ mW,
                                                   float* __restrict__ vW, float* __restrict__ bias, float* __restrict__ mb,
                                                   float* __restrict__ vb, AdamC ad) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    auto a = [=] __device__(int m, int k) -> float {
        const float v = L[(size_t)k * Min + min(m, Min - 1)];
        return m < Min ? v : 1.f;
    };
    auto b = [=] __device__(int k, int n) -> float { return R[(size_t)k * N + n]; };
    auto epi = [=] __device__(int m, int n, float g) {
        if (m < Min) adam_update(W, mW, vW, (size_t)m * N + n, g, ad);
        else adam_update(bias, mb, vb, n, g, ad);
    };
    if constexpr (V) {   // Min % 4 == 0: the ones row (m == Min) opens its own group
        auto a4 = [=] __device__(int m, int k) -> float4 {
            float4 v = ltg_ld4(L + (size_t)min(k, K - 1) * Min, m, Min, k < K);
            if (m == Min && k < K) v.x = 1.f;
            return v;
        };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(R + (size_t)min(k, K - 1) * N, n, N, k < K); };
        ltg_gemm_block<false, 32, 32, 128, 2, 2, true, true, false, 0, 0, 3>(Min + 1, N, m0, n0, 0, K, a4, b4, epi);
    } else {
        ltg_gemm_block<false, 32, 32, 128, 2, 2, true, true>(Min + 1, N, m0, n0, 0, K, a, b, epi);
    }
}

// dh1 = dmlv . W_q1^T ; da1 = dh1 * (1 - h1^2)
template <bool V, int BKV = 128>
__global__ __launch_bounds__(NT) void k_dh1(int B, int H, int Z2, const float* __restrict__ dmlv,
                                            const float* __restrict__ Wq1, const float* __restrict__ h1, float* __restrict__ da1) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    auto a = [=] __device__(int m, int k) -> float { return dmlv[(size_t)m * Z2 + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Wq1[(size_t)n * Z2 + k]; };
    auto epi = [=] __device__(int m, int n, float acc) {
        const float t = h1[(size_t)m * H + n];
        da1[(size_t)m * H + n] = acc * (1.f - t * t);
    };
    if constexpr (V) {
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(dmlv + (size_t)min(m, B - 1) * Z2, k, Z2, m < B); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Wq1 + (size_t)min(n, H - 1) * Z2, k, Z2, n < H); };
        ltg_gemm_block<false, 32, 32, BKV, 2, 2, false, false, false, 0, 0, 3>(B, H, m0, n0, 0, Z2, a4, b4, epi);
    } else {
        ltg_gemm_block<false, 32, 32, 128, 2, 2, false, false>(B, H, m0, n0, 0, Z2, a, b, epi);
    }
}

// Sparse gradient rows of W_q0: G[u][:] = sum over the batch entries of item uitem[u] of
// keep * val * row_scale[b] * da1[b][:]; row n_unique = bias gradient sum_b da1[b][:].
// One workgroup per distinct item: the 4 waves split its entries (a popular item is in dozens of the
// batch's rows), lanes own float4 column chunks, partials meet in LDS.
constexpr int ENC0_BIAS_PARTS = 8;  // the bias gradient (column sum of da1 over the batch) is cut into this many partial rows
__global__ __launch_bounds__(NT) void k_enc0_grad(int B, int I, int H, int nu, const int32_t* __restrict__ uptr,
                                                  const int32_t* __restrict__ rowidx, const int32_t* __restrict__ csr_pos,
                                                  const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                  const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed, uint64_t step,
                                                  const float* __restrict__ row_scale, const float* __restrict__ da1,
                                                  float* __restrict__ G, int item_lo, int Ig) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];  // [4][H]
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H4 = H >> 2;
    constexpr int MAXQ = 4;
    float4 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* d4 = reinterpret_cast<const float4*>(da1);
    // workgroups u < nu: one distinct item each; u >= nu: part (u - nu) of the bias row = batch rows [q0, q1) (one long row
    // of B entries would be the launch's critical path)
    const int bp = u - nu, per = (B + ENC0_BIAS_PARTS - 1) / ENC0_BIAS_PARTS;
    const int q0 = u < nu ? uptr[u] : min(B, bp * per), q1 = u < nu ? uptr[u + 1] : min(B, (bp + 1) * per);
    // 4 entries per trip and wave: their (dependent) index chains and da1 row loads overlap
    for (int q = q0 + w; q < q1; q += 4 * (NT / 64)) {
        int b[4];
        float sc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int qt = q + t * (NT / 64);
            const bool ok = qt < q1;
            const int qc = ok ? qt : q;
            if (u < nu) {
                b[t] = rowidx[qc];
                const int pos = csr_pos[qc];
                const int it = indices[pos];
                const bool kp = drop_keep ? (drop_keep[pos] != 0)
                                          : ltg_rng_keep(seed, LTG_STREAM_VAE_DROPOUT, step, (uint64_t)b[t] * (uint64_t)Ig + item_lo + it, keep);
                sc[t] = (ok && kp) ? (values ? values[pos] : 1.f) * row_scale[b[t]] : 0.f;
            } else {
                b[t] = qc;  // bias row: every batch row, weight 1
                sc[t] = ok ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int qq = 0; qq < MAXQ; ++qq) {
            const int c4 = lane + 64 * qq;
            if (c4 < H4) {
                float4 d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) d[t] = d4[(size_t)b[t] * H4 + c4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[qq].x += sc[t] * d[t].x;
                    acc[qq].y += sc[t] * d[t].y;
                    acc[qq].z += sc[t] * d[t].z;
                    acc[qq].w += sc[t] * d[t].w;
                }
            }
        }
    }
#pragma unroll
    for (int qq = 0; qq < MAXQ; ++qq) {
        const int c4 = lane + 64 * qq;
        if (c4 < H4) reinterpret_cast<float4*>(s_g + (size_t)w * H)[c4] = acc[qq];
    }
    __syncthreads();
    for (int c = tid; c < H; c += NT) G[(size_t)u * H + c] = s_g[c] + s_g[H + c] + s_g[2 * H + c] + s_g[3 * H + c];
}

// Dense Adam sweep over W_q0 [I][H] (+ bias row I): pure streaming, 16 B per lane, the sparse gradient row
// (if any) is picked up through slot[i].  TF's Adam touches every row every step (a zero gradient still
// decays m, v and moves theta), so this sweep is the algorithmic 24 B/parameter.
// item -> gradient row map of ONE batch, built on the fly when the caller keeps no per-batch slot[] cache (ltg_batch.slot ==
// NULL): map[] was memset to -1; group u's item id is the column of its first entry
__global__ __launch_bounds__(NT) void k_fill_i32(int n, int32_t v, int32_t* __restrict__ p) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) p[i] = v;
}
__global__ __launch_bounds__(NT) void k_slot_scatter(int nu, const int32_t* __restrict__ uptr, const int32_t* __restrict__ csr_pos,
                                                     const int32_t* __restrict__ indices, int32_t* __restrict__ map) {
    const int u = blockIdx.x * NT + threadIdx.x;
    if (u < nu) map[indices[csr_pos[uptr[u]]]] = u;
}

__global__ __launch_bounds__(NT) void k_enc0_bwd_adam(int I, int H, int nu, const int32_t* __restrict__ slot,
                                                      const float* __restrict__ G, ltg_gen_state st, AdamC ad) {
    const int H4 = H >> 2;  // H % 4 == 0 (checked on the host)
    const size_t total = (size_t)(I + 1) * H4;
    float4* W4 = reinterpret_cast<float4*>(st.p[0]);
    float4* m4 = reinterpret_cast<float4*>(st.m[0]);
    float4* v4 = reinterpret_cast<float4*>(st.v[0]);
    float4* b4 = reinterpret_cast<float4*>(st.p[4]);
    float4* mb4 = reinterpret_cast<float4*>(st.m[4]);
    float4* vb4 = reinterpret_cast<float4*>(st.v[4]);
    const float4* G4 = reinterpret_cast<const float4*>(G);
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const int i = (int)(e / H4), c = (int)(e % H4);
        float4* P = i < I ? W4 + e : b4 + c;
        float4* Mm = i < I ? m4 + e : mb4 + c;
        float4* Vv = i < I ? v4 + e : vb4 + c;
        float4 p = *P, mm = *Mm, vv = *Vv;
        const int u = i < I ? slot[i] : nu;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u >= 0) g = G4[(size_t)u * H4 + c];
        if (i >= I) {   // bias row: the remaining partial rows of k_enc0_grad
#pragma unroll
            for (int j = 1; j < ENC0_BIAS_PARTS; ++j) {
                const float4 t = G4[(size_t)(nu + j) * H4 + c];
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
}
