// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): fake-pair sampler (sample.py:40-67, train.py:227-251) and ranking metrics (eval_functions.py:11-62).
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------
// Sampler: sample_from_generator_new (sample.py:40-67) + pair construction (train.py:227-251).
// One wave per user.  Successive sampling without replacement == Gumbel-top-k on log p.
// ---------------------------------------------------------------------------------------------
constexpr int SP_NT = 1024;   // 16 waves: the rank loop is arithmetic over LDS broadcasts -- four waves per SIMD hide the LDS latency
__global__ __launch_bounds__(SP_NT) void k_sample_pairs(int I, const int32_t* __restrict__ cand_ptr,
                                                        const int32_t* __restrict__ cand_idx, const int32_t* __restrict__ pop_ptr,
                                                        const int32_t* __restrict__ pop_idx, const int32_t* __restrict__ n_sample,
                                                        const int32_t* __restrict__ slot_ptr, const uint8_t* __restrict__ valid_item,
                                                        const float* __restrict__ u_gumbel, const float* __restrict__ u_pick,
                                                        uint64_t seed, uint64_t step, const float* __restrict__ logits,
                                                        const float* __restrict__ lse, int32_t* __restrict__ gen_out,
                                                        int32_t* __restrict__ pop_out, int32_t* __restrict__ cnt_out,
                                                        const float* __restrict__ cand_logit, int rps) {
    // I is the GLOBAL item count (RNG index space); cand_logit (optional, aligned with cand_idx) replaces
    // the [B, I] logits matrix when the items are sharded over ranks.
    extern __shared__ __attribute__((aligned(16))) float s_key[];
    __shared__ int s_w[SP_NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ns = n_sample[b];
    const int s0 = slot_ptr[b];
    if (ns <= 0) return;  // uniform for the whole workgroup
    // several batches in one launch (ltg_sample_inputs.rows_per_step): the row's own batch counter, its row there, its batch's count
    const uint64_t kb = rps > 0 ? (uint64_t)(b % rps) : (uint64_t)b;
    step += rps > 0 ? (uint64_t)(b / rps) : 0;
    cnt_out += rps > 0 ? b / rps : 0;
    const int c0 = cand_ptr[b], nc = cand_ptr[b + 1] - c0;
    const float l = lse[b];
    const float* row = logits + (size_t)b * I;
    // sum over the workgroup of a small non-negative count (all threads get it)
    auto block_count = [&](int x) -> int {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        __syncthreads();
        if (lane == 0) s_w[w] = x;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int i = 0; i < SP_NT / 64; ++i) t += s_w[i];
        return t;
    };
    int nnz_l = 0;
    for (int j = tid; j < nc; j += SP_NT) {
        const int it = cand_idx[c0 + j];
        const float lp = (cand_logit ? cand_logit[c0 + j] : row[it]) - l;
        const bool pos = expf(lp) > 0.f;  // softmax underflow == "zero probability" of sample.py:45
        float u = u_gumbel ? u_gumbel[c0 + j] : ltg_rng_uniform(seed, LTG_STREAM_GUMBEL, step, kb * (uint64_t)I + it);
        u = fmaxf(u, 2.98023223876953125e-8f);  // 2^-25
        s_key[j] = pos ? lp - logf(-logf(u)) : -INFINITY;
        nnz_l += pos ? 1 : 0;
    }
    const int nnz = block_count(nnz_l);  // includes the barrier that publishes s_key
    const int k_eff = min(ns, nnz);  // Q10: exception-driven decrement of to_sample
    const int np = pop_ptr[b + 1] - pop_ptr[b];
    // Selected = the k_eff largest keys, ties to the smaller index.  Fast pass: g(j) = #{t : key_t > key_j} (one compare per
    // pair, 16-byte LDS broadcasts).  {j : g(j) < k_eff} is the selected set unless equal keys straddle the boundary -- then it
    // is larger than k_eff, and the exact ranks (with the index tie-break) are counted instead.  A -inf key has g >= nnz >= k_eff.
    const int nc4 = nc >> 2;
    const ltg_f32x4* k4 = reinterpret_cast<const ltg_f32x4*>(s_key);
    unsigned selmask = 0;   // bit p: candidate j = tid + p * SP_NT (<= 16 passes: max_cand <= 16384)
    int nsel_l = 0;
    for (int j = tid, p = 0; j < nc; j += SP_NT, ++p) {
        const float kj = s_key[j];
        int g = 0;
        for (int t = 0; t < nc4; ++t) {
            const ltg_f32x4 k = k4[t];
            g += (k[0] > kj ? 1 : 0) + (k[1] > kj ? 1 : 0) + (k[2] > kj ? 1 : 0) + (k[3] > kj ? 1 : 0);
        }
        for (int t = nc4 * 4; t < nc; ++t) g += s_key[t] > kj ? 1 : 0;
        if (g < k_eff) { selmask |= 1u << p; ++nsel_l; }
    }
    if (block_count(nsel_l) != k_eff) {   // equal keys at the boundary (uniform branch)
        selmask = 0;
        for (int j = tid, p = 0; j < nc; j += SP_NT, ++p) {
            const float kj = s_key[j];
            int rank = 0;
            for (int t = 0; t < nc; ++t) {
                const float kt = s_key[t];
                rank += (kt > kj || (kt == kj && t < j)) ? 1 : 0;
            }
            if (rank < k_eff) selmask |= 1u << p;
        }
    }
    int written = 0;
    int ok_l = 0;
    for (int j0 = 0, p = 0; j0 < nc; j0 += SP_NT, ++p) {
        const int j = j0 + tid;
        const bool sel = (selmask >> p) & 1u;
        const unsigned long long bal = __ballot(sel);
        __syncthreads();
        if (lane == 0) s_w[w] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < SP_NT / 64; ++i) {
            before += i < w ? s_w[i] : 0;
            total += s_w[i];
        }
        if (sel) {
            const int pos = written + before + __popcll(bal & ((1ull << lane) - 1ull));
            const int s = s0 + pos;
            const int gid = cand_idx[c0 + j];
            const float u = u_pick ? u_pick[s] : ltg_rng_uniform(seed, LTG_STREAM_POP_PICK, step, kb * (uint64_t)I + gid);
            const int pi = min((int)(u * (float)np), np - 1);  // np.random.choice(range(n)) train.py:236
            const int pid = pop_idx[pop_ptr[b] + pi];
            const bool ok = valid_item[gid] != 0 && valid_item[pid] != 0;  // train.py:240
            gen_out[s] = ok ? gid : -1;
            pop_out[s] = ok ? pid : -1;
            ok_l += ok ? 1 : 0;
        }
        written += total;
    }
    for (int s = s0 + written + tid; s < s0 + ns; s += SP_NT) {
        gen_out[s] = -1;
        pop_out[s] = -1;
    }
    const int okcnt = block_count(ok_l);
    if (tid == 0 && okcnt > 0) atomicAdd(cnt_out, okcnt);
}

// ---------------------------------------------------------------------------------------------
// Ranking metrics (eval_functions.py:11-62, train.py:341).  One workgroup per user.
// rank(h) = #{i : score_i > score_h or (score_i == score_h and i < h)}, score = -inf on fold-in items.
// ---------------------------------------------------------------------------------------------
constexpr int RM_T = 16;  // held-out items processed per pass
// Shared by the one-GPU path (score_in == nullptr, count_out == nullptr: everything in one launch) and the item-sharded
// path (this rank's slab [item_lo, item_lo + I): scores of the held-out entries come all-reduced in score_in, the counts
// of LOCAL items that beat each entry go to count_out for the all-reduce; te ids are GLOBAL, tr ids LOCAL).
__device__ __forceinline__ void rank_finish_row(const int* cnt, int np, int k_ndcg, int k_r1, int k_r2, double* acc) {
    for (int t = 0; t < np; ++t) {
        const int r = cnt[t];
        if (r < k_ndcg) acc[0] += 1.0 / log2((double)r + 2.0);
        if (r < k_r1) acc[1] += 1.0;
        if (r < k_r2) acc[2] += 1.0;
    }
}
__device__ __forceinline__ void rank_write_row(float* out, const double* acc, int nte, int k_ndcg, int k_r1, int k_r2) {
    double idcg = 0.0;
    for (int r = 0; r < min(nte, k_ndcg); ++r) idcg += 1.0 / log2((double)r + 2.0);
    out[0] = idcg != 0.0 ? (float)(acc[0] / idcg) : 0.f;
    out[1] = nte > 0 ? (float)(acc[1] / (double)min(k_r1, nte)) : 0.f;
    out[2] = nte > 0 ? (float)(acc[2] / (double)min(k_r2, nte)) : 0.f;
    out[3] = idcg != 0.0 ? 1.f : 0.f;
}

__global__ __launch_bounds__(NT) void k_rank_metrics(int I, int item_lo, const float* __restrict__ logits, const int32_t* __restrict__ tr_ptr,
                                                     const int32_t* __restrict__ tr_idx, const int32_t* __restrict__ te_ptr,
                                                     const int32_t* __restrict__ te_idx, const float* __restrict__ score_in,
                                                     int32_t* __restrict__ count_out, int k_ndcg, int k_r1, int k_r2,
                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_bits[];  // ceil(I/32) words
    __shared__ int s_cnt[RM_T];
    __shared__ float s_sc[RM_T];
    __shared__ int s_it[RM_T];
    __shared__ double s_acc[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nw = (I + 31) >> 5;
    for (int w = tid; w < nw; w += NT) s_bits[w] = 0u;
    if (tid < 4) s_acc[tid] = 0.0;
    __syncthreads();
    for (int e = tr_ptr[b] + tid; e < tr_ptr[b + 1]; e += NT) {
        const int it = tr_idx[e];
        atomicOr(&s_bits[it >> 5], 1u << (it & 31));
    }
    __syncthreads();
    const float* row = logits + (size_t)b * I;
    const int t0 = te_ptr[b], nte = te_ptr[b + 1] - t0;
    for (int p0 = 0; p0 < nte; p0 += RM_T) {
        const int np = min(RM_T, nte - p0);
        if (tid < np) {
            const int it = te_idx[t0 + p0 + tid];       // global id
            s_it[tid] = it;
            if (score_in) {
                s_sc[tid] = score_in[t0 + p0 + tid];
            } else {
                const int l = it - item_lo;
                s_sc[tid] = ((s_bits[l >> 5] >> (l & 31)) & 1u) ? -INFINITY : row[l];
            }
            s_cnt[tid] = 0;
        }
        __syncthreads();
        int cnt[RM_T];
#pragma unroll
        for (int t = 0; t < RM_T; ++t) cnt[t] = 0;
        for (int i = tid; i < I; i += NT) {
            const float sc = ((s_bits[i >> 5] >> (i & 31)) & 1u) ? -INFINITY : row[i];
            const int ig = i + item_lo;
#pragma unroll
            for (int t = 0; t < RM_T; ++t)
                if (t < np) cnt[t] += (sc > s_sc[t] || (sc == s_sc[t] && ig < s_it[t])) ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < RM_T; ++t) {
            if (t < np) {
                int c = cnt[t];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                if ((tid & 63) == 0) atomicAdd(&s_cnt[t], c);
            }
        }
        __syncthreads();
        if (count_out) {
            if (tid < np) count_out[t0 + p0 + tid] = s_cnt[tid];
        } else if (tid == 0) {
            rank_finish_row(s_cnt, np, k_ndcg, k_r1, k_r2, s_acc);
        }
        __syncthreads();
    }
    if (tid == 0 && !count_out) rank_write_row(out + (size_t)b * 4, s_acc, nte, k_ndcg, k_r1, k_r2);
}

// scores of the held-out entries this rank owns (-inf on fold-in items), 0 for the others -> all-reduce(sum)
__global__ __launch_bounds__(NT) void k_rank_scores(int I, int item_lo, int n_rows, const float* __restrict__ logits,
                                                    const int32_t* __restrict__ tr_ptr, const int32_t* __restrict__ tr_idx,
                                                    const int32_t* __restrict__ te_ptr, const int32_t* __restrict__ te_idx,
                                                    float* __restrict__ score_out) {
    const int b = blockIdx.x;
    const int a0 = tr_ptr[b], a1 = tr_ptr[b + 1];
    for (int e = te_ptr[b] + threadIdx.x; e < te_ptr[b + 1]; e += NT) {
        const int l = te_idx[e] - item_lo;
        float sc = 0.f;
        if (l >= 0 && l < I) {
            int lo = a0, hi = a1;                       // tr rows are sorted (CSR with sorted indices)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (tr_idx[mid] < l) lo = mid + 1; else hi = mid;
            }
            sc = (lo < a1 && tr_idx[lo] == l) ? -INFINITY : logits[(size_t)b * I + l];
        }
        score_out[e] = sc;
    }
}

__global__ void k_rank_finish(int n_rows, const int32_t* __restrict__ te_ptr, const int32_t* __restrict__ counts, int k_ndcg, int k_r1,
                              int k_r2, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    double acc[3] = {0.0, 0.0, 0.0};
    const int t0 = te_ptr[b], nte = te_ptr[b + 1] - t0;
    rank_finish_row(counts + t0, nte, k_ndcg, k_r1, k_r2, acc);
    rank_write_row(out + (size_t)b * 4, acc, nte, k_ndcg, k_r1, k_r2);
}
