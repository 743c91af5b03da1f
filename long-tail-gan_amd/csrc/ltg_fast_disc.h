// Part of csrc/ltg_fast.h (included there, in this order, inside ltg_kernels.hip's anonymous namespace): latency-path kernels of the fp32 discriminator step at config.ini-sized layers (fk_d_l1, fk_d_l2, fk_d_y, fk_d_bwd1, fk_d_bwd2, fk_d_adam); d_arith selects their product form.
// Split out of the 2 100-line header in round 6 -- the code is unchanged.
#pragma once

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
