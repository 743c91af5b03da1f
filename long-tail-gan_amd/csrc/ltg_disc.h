// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): PairView / DropView and the generic (LDS-staged, any size, fp32 / bf16 / fp8) discriminator kernels (discriminator.py:3-58).
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------
// Discriminator (discriminator.py:3-58).  A "pair batch" is the logical concatenation of the real
// tower rows [0, nr) and the fake tower rows [nr, nr+nf); both towers share the weights.
// ---------------------------------------------------------------------------------------------
struct PairView {
    int nr, nf;
    const int32_t *r_pop, *r_nic, *f_pop, *f_nic;
    // pointer/index selects + ONE unconditional load (no divergent branch around the load)
    __device__ __forceinline__ int pop(int r) const {
        const int32_t* p = r < nr ? r_pop : f_pop;
        return p[r < nr ? r : r - nr];
    }
    __device__ __forceinline__ int nic(int r) const {
        const int32_t* p = r < nr ? r_nic : f_nic;
        return p[r < nr ? r : r - nr];
    }
    // (both ids requested unconditionally: with `&&` the niche id was loaded only under the popular id's sign -- a dependent round trip)
    __device__ __forceinline__ bool valid(int r) const { return (pop(r) | nic(r)) >= 0; }
};
struct DropView {
    const uint8_t *real, *fake;  // optional injected keep flags [rows][width]
    int nr;
    int row0;                    // logical pair row of local row 0 (a rank that owns rows [row0, ...) of the pair batch draws the
                                 // mask the whole batch draws: the counter RNG is indexed by the GLOBAL row)
    // several pair batches in one pass (ltg_fake_tower_batched): row r belongs to batch seg_of[r], which starts at row seg_row0[.]
    // and draws with counter seg_step[.]
    const int32_t *seg_of = nullptr, *seg_row0 = nullptr;
    const uint64_t* seg_step = nullptr;
    __device__ __forceinline__ bool keep(int r, int c, int width, uint64_t seed, uint32_t stream, uint64_t step, float kp) const {
        if (real || fake) return r < nr ? (real[(size_t)r * width + c] != 0) : (fake[(size_t)(r - nr) * width + c] != 0);
        if (seg_of) {
            const int sg = seg_of[r];
            return ltg_rng_keep(seed, stream, seg_step[sg], (uint64_t)(r - seg_row0[sg]) * width + c, kp);
        }
        return ltg_rng_keep(seed, stream, step, (uint64_t)(r + row0) * width + c, kp);
    }
};

// Discriminator GEMM precision (ltg_config.d_precision): 0 = fp32 MFMA (the reference's arithmetic), 1 = bf16 operands,
// 2 = OCP e4m3 operands with STATIC power-of-two scales per operand class (no amax pass: the classes are bounded --
// embeddings and weights are N(0, 0.1) truncated at 2 sigma at initialisation, activations are tanh / keep, the gradient
// classes are bounded by products of those); accumulation is fp32 in every mode.  TS = tile size (32: latency-bound
// default sizes; 128 = the wide discriminator of BASELINE config 5: 128x128x64 tiles fed by 16-B vector loads, every
// dimension a multiple of 4).
constexpr int FP8_S_EMB = 8, FP8_S_W = 8, FP8_S_ACT = 6, FP8_S_G3 = 8, FP8_S_G1 = 7;
// branch layers (discriminator.py:16-19,25,30,51,52): blockIdx.z = 0 popular->h1, 1 niche->h2
template <int MODE, int TS, int V>
__global__ __launch_bounds__(NT) void k_d_l1(PairView pv, int h0, int h1, int h2, const float* __restrict__ emb,
                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2, DropView dA,
                                             DropView dB, float keep, uint64_t seed, uint64_t step,
                                             float* __restrict__ A1) {
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const bool br = blockIdx.z != 0;
    const int N = br ? h2 : h1;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    if (n0 >= N) return;
    const float* W = br ? w2 : w1;
    const float* bias = br ? b2 : b1;
    auto a = [=] __device__(int m, int k) -> float {
        const int id = br ? pv.nic(m) : pv.pop(m);
        const float v = emb[(size_t)max(id, 0) * h0 + k];
        return id >= 0 ? v : 0.f;
    };
    auto b = [=] __device__(int k, int nn) -> float { return W[(size_t)k * N + nn]; };
    auto epi = [=] __device__(int m, int nn, float acc) {
        const float t = tanhf(acc + bias[nn]);
        const bool kp = br ? dB.keep(m, nn, h2, seed, LTG_STREAM_D_DROP_B, step, keep)
                           : dA.keep(m, nn, h1, seed, LTG_STREAM_D_DROP_A, step, keep);
        A1[(size_t)m * h12 + (br ? h1 : 0) + nn] = kp ? t / keep : 0.f;
    };
    if constexpr (V) {
        auto a4 = [=] __device__(int m, int k) -> float4 {
            const int id = br ? pv.nic(min(m, n - 1)) : pv.pop(min(m, n - 1));
            return ltg_ld4(emb + (size_t)max(id, 0) * h0, k, h0, m < n && id >= 0);
        };
        auto b4 = [=] __device__(int k, int nn) -> float4 { return ltg_ld4(W + (size_t)min(k, h0 - 1) * N, nn, N, k < h0); };
        if constexpr (V == 3) ltg_gemm_block<MODE, TS, TS, (TS == 32 ? 128 : 64), 2, 2, false, true, false, FP8_S_EMB, FP8_S_W, 3>(n, N, m0, n0, 0, h0, a4, b4, epi);
        else ltg_gemm_block<MODE, TS, TS, 128, 2, 2, false, true, false, FP8_S_EMB, FP8_S_W, 1>(n, N, m0, n0, 0, h0, a4, b, epi);   // h1 / h2 not multiples of 4
    } else {
        ltg_gemm_block<MODE, TS, TS, 128, 2, 2, false, true, false, FP8_S_EMB, FP8_S_W>(n, N, m0, n0, 0, h0, a, b, epi);
    }
}

// fully connected layer (discriminator.py:44, :54)
template <int MODE, int TS, int V>
__global__ __launch_bounds__(NT) void k_d_l2(int n, int h12, int h3, const float* __restrict__ A1,
                                             const float* __restrict__ w3, const float* __restrict__ b3, DropView dC,
                                             float keep, uint64_t seed, uint64_t step, float* __restrict__ A3) {
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    auto a = [=] __device__(int m, int k) -> float { return A1[(size_t)m * h12 + k]; };
    auto b = [=] __device__(int k, int nn) -> float { return w3[(size_t)k * h3 + nn]; };
    auto epi = [=] __device__(int m, int nn, float acc) {
        const float t = tanhf(acc + b3[nn]);
        A3[(size_t)m * h3 + nn] = dC.keep(m, nn, h3, seed, LTG_STREAM_D_DROP_C, step, keep) ? t / keep : 0.f;
    };
    if constexpr (V) {
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(A1 + (size_t)min(m, n - 1) * h12, k, h12, m < n); };
        auto b4 = [=] __device__(int k, int nn) -> float4 { return ltg_ld4(w3 + (size_t)min(k, h12 - 1) * h3, nn, h3, k < h12); };
        ltg_gemm_block<MODE, TS, TS, (TS == 32 ? 128 : 64), 2, 2, false, true, false, FP8_S_ACT, FP8_S_W, 3>(n, h3, m0, n0, 0, h12, a4, b4, epi);
    } else {
        ltg_gemm_block<MODE, TS, TS, 128, 2, 2, false, true, false, FP8_S_ACT, FP8_S_W>(n, h3, m0, n0, 0, h12, a, b, epi);
    }
}

// output unit + loss terms (discriminator.py:45,55; train.py:142): one wave per pair row.
// y[r] (0 for holes), ds[r] = d d_loss / d s_r, lrow[r] = loss term, and (WITH_BWD) the gradient at
// the fc layer's pre-activation dpre3[r][c] = ds * w4[c] * dact(A3[r][c]) for the backward GEMMs.
template <bool WITH_BWD>
__global__ __launch_bounds__(NT) void k_d_out(PairView pv, int h3, const float* __restrict__ A3,
                                              const float* __restrict__ w4, const float* __restrict__ b4, float keep,
                                              float* __restrict__ y, float* __restrict__ ds, float* __restrict__ lrow,
                                              float* __restrict__ dpre3) {
    const int n = pv.nr + pv.nf;
    const int r = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < h3; c += 64) s += A3[(size_t)r * h3 + c] * w4[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s += b4[0];
    const float yy = 1.f / (1.f + expf(-s));
    const bool ok = pv.valid(r);
    const bool real = r < pv.nr;
    const float dsr = ok ? (real ? -(1.f - yy) : yy) : 0.f;
    if (lane == 0) {
        y[r] = ok ? yy : 0.f;
        ds[r] = dsr;
        lrow[r] = ok ? (real ? -logf(yy) : -logf(1.f - yy)) : 0.f;
    }
    if (WITH_BWD) {
        const float ik = 1.f / keep;
        for (int c = lane; c < h3; c += 64) {
            const float a = A3[(size_t)r * h3 + c];
            const float t = a * keep;
            dpre3[(size_t)r * h3 + c] = a != 0.f ? dsr * w4[c] * (1.f - t * t) * ik : 0.f;
        }
    }
}

// derivative through dropout(tanh(.)): a = t/keep*mask  =>  d pre = d a * (1 - t^2)/keep where mask=1
__device__ __forceinline__ float dact(float a, float keep) {
    const float t = a * keep;
    return a != 0.f ? (1.f - t * t) / keep : 0.f;
}

// Flat layout of the discriminator's trainable tensors (discriminator.py:47 order) used by the
// split-K gradient slabs and the single Adam sweep.
struct DLayout {
    int off[9];  // off[i] = start of tensor i, off[8] = total
};
__host__ __device__ inline DLayout d_layout(int h0, int h1, int h2, int h3) {
    DLayout L;
    const int sz[8] = {h0 * h1, h1, h0 * h2, h2, (h1 + h2) * h3, h3, h3, 1};
    L.off[0] = 0;
    for (int i = 0; i < 8; ++i) L.off[i + 1] = L.off[i] + sz[i];
    return L;
}
constexpr int D_KCHUNK = 256;  // pair rows per split-K slab

// Backward stage 1, ONE launch, three independent jobs selected by the block index:
//   job A  dpre1 = (dpre3 . w3^T) * dact(A1)                       [n][h1+h2]     tiles 64x64
//   job B  slab[z] += A1^T . dpre3 (+ ones row -> db3), split-K     [(h12+1)][h3]  tiles 32x32
//   job C  slab[z] += A3^T . ds, sum ds (dw4, db4), split-K         column reduce
template <int MODE, int TS, int V>
__global__ __launch_bounds__(NT) void k_d_bwd1(int n, int h12, int h3, int nA, int nB, int ks, DLayout L,
                                               const float* __restrict__ A1, const float* __restrict__ A3,
                                               const float* __restrict__ ds, const float* __restrict__ dpre3,
                                               const float* __restrict__ w3, float keep, float* __restrict__ dpre1,
                                               float* __restrict__ slab) {
    int bid = blockIdx.x;
    if (bid < nA) {
        const int tn = (h12 + TS - 1) / TS;
        const int m0 = (bid / tn) * TS, n0 = (bid % tn) * TS;
        auto a = [=] __device__(int m, int k) -> float { return dpre3[(size_t)m * h3 + k]; };
        auto b = [=] __device__(int k, int nn) -> float { return w3[(size_t)nn * h3 + k]; };
        auto epi = [=] __device__(int m, int nn, float acc) {
            // (MODE 2, e4m3 operands: the derivative rounded to bf16 -- the form the operand-format backward stores it in (ltg_fp8bwd.h:
            // dA1T_16), so that both fp8 paths feed the e4m3 conversion of dpre1 the same values)
            const float da = dact(A1[(size_t)m * h12 + nn], keep);
            dpre1[(size_t)m * h12 + nn] = acc * (MODE == 2 ? __uint_as_float((unsigned)ltg_f2bf(da) << 16) : da);
        };
        if constexpr (V) {
            auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(dpre3 + (size_t)min(m, n - 1) * h3, k, h3, m < n); };
            auto b4 = [=] __device__(int k, int nn) -> float4 { return ltg_ld4(w3 + (size_t)min(nn, h12 - 1) * h3, k, h3, nn < h12); };
            ltg_gemm_block<MODE, TS, TS, (TS == 32 ? 128 : 64), 2, 2, false, false, false, FP8_S_G3, FP8_S_W, 3>(n, h12, m0, n0, 0, h3, a4, b4, epi);
        } else {
            ltg_gemm_block<MODE, TS, TS, 128, 2, 2, false, false, false, FP8_S_G3, FP8_S_W>(n, h12, m0, n0, 0, h3, a, b, epi);
        }
        return;
    }
    bid -= nA;
    const int P = L.off[8];
    if (bid < nB) {
        const int tm = (h12 + 1 + TS - 1) / TS, tn = (h3 + TS - 1) / TS;
        const int z = bid / (tm * tn), t = bid % (tm * tn);
        const int m0 = (t / tn) * TS, n0 = (t % tn) * TS;
        const int kbeg = z * D_KCHUNK, kend = min(n, kbeg + D_KCHUNK);
        float* out = slab + (size_t)z * P;
        const int ow = L.off[4], ob = L.off[5];
        auto a = [=] __device__(int m, int k) -> float {
            const float v = A1[(size_t)k * h12 + min(m, h12 - 1)];
            return m < h12 ? v : 1.f;
        };
        auto b = [=] __device__(int k, int nn) -> float { return dpre3[(size_t)k * h3 + nn]; };
        auto epi = [=] __device__(int m, int nn, float g) {
            if (m < h12) out[ow + (size_t)m * h3 + nn] = g;
            else out[ob + nn] = g;
        };
        if constexpr (V) {
            // rows m < h12: A1^T; row m == h12: ones (bias gradient); h12 % 4 == 0, so the ones row opens its own group
            auto a4 = [=] __device__(int m, int k) -> float4 {
                float4 v = ltg_ld4(A1 + (size_t)min(k, kend - 1) * h12, m, h12, k < kend);
                if (m == h12 && k < kend) v.x = 1.f;
                return v;
            };
            auto b4 = [=] __device__(int k, int nn) -> float4 { return ltg_ld4(dpre3 + (size_t)min(k, kend - 1) * h3, nn, h3, k < kend); };
            ltg_gemm_block<MODE, TS, TS, (TS == 32 ? 128 : 64), 2, 2, true, true, false, FP8_S_ACT, FP8_S_G3, 3>(h12 + 1, h3, m0, n0, kbeg, kend, a4, b4, epi);
        } else {
            ltg_gemm_block<MODE, TS, TS, 128, 2, 2, true, true, false, FP8_S_ACT, FP8_S_G3>(h12 + 1, h3, m0, n0, kbeg, kend, a, b, epi);
        }
        return;
    }
    bid -= nB;
    {
        __shared__ float part[8][33];
        const int tc = (h3 + 1 + 31) / 32;
        const int z = bid / tc;
        const int tn = threadIdx.x & 31, tr = threadIdx.x >> 5;
        const int c = (bid % tc) * 32 + tn;  // c == h3 is the bias column
        const int kbeg = z * D_KCHUNK, kend = min(n, kbeg + D_KCHUNK);
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
            slab[(size_t)z * P + (c < h3 ? L.off[6] + c : L.off[7])] = g;
        }
    }
}

// Backward stage 2: dw1/db1 and dw2/db2 slabs (E_pop^T . dpre1[:, :h1], E_niche^T . dpre1[:, h1:]), split-K.
template <int MODE, int TS, int V>
__global__ __launch_bounds__(NT) void k_d_bwd2(PairView pv, int h0, int h1, int h2, int ks, DLayout L,
                                               const float* __restrict__ emb, const float* __restrict__ dpre1,
                                               float* __restrict__ slab) {
    const int n = pv.nr + pv.nf, h12 = h1 + h2;
    const int tm = (h0 + 1 + TS - 1) / TS;
    const int tn1 = (h1 + TS - 1) / TS, tn2 = (h2 + TS - 1) / TS;
    const int per_z = tm * (tn1 + tn2);
    const int z = blockIdx.x / per_z, t = blockIdx.x % per_z;
    const int m0 = (t / (tn1 + tn2)) * TS;
    const int tcol = t % (tn1 + tn2);
    const bool br = tcol >= tn1;
    const int n0 = (br ? tcol - tn1 : tcol) * TS;
    const int N = br ? h2 : h1;
    const int coff = br ? h1 : 0;
    const int ow = L.off[br ? 2 : 0], ob = L.off[br ? 3 : 1];
    const int kbeg = z * D_KCHUNK, kend = min(n, kbeg + D_KCHUNK);
    float* out = slab + (size_t)z * L.off[8];
    auto a = [=] __device__(int m, int k) -> float {
        const int id = br ? pv.nic(k) : pv.pop(k);
        const float v = emb[(size_t)max(id, 0) * h0 + min(m, h0 - 1)];
        return m < h0 ? (id >= 0 ? v : 0.f) : 1.f;
    };
    auto b = [=] __device__(int k, int nn) -> float { return dpre1[(size_t)k * h12 + coff + nn]; };
    auto epi = [=] __device__(int m, int nn, float g) {
        if (m < h0) out[ow + (size_t)m * N + nn] = g;
        else out[ob + nn] = g;
    };
    if constexpr (V) {
        auto a4 = [=] __device__(int m, int k) -> float4 {
            const int kc = min(k, kend - 1);
            const int id = br ? pv.nic(kc) : pv.pop(kc);
            float4 v = ltg_ld4(emb + (size_t)max(id, 0) * h0, m, h0, k < kend && id >= 0);
            if (m == h0 && k < kend) v.x = 1.f;       // the ones row: bias gradient (also for pairs with a hole: like the scalar path)
            return v;
        };
        auto b4 = [=] __device__(int k, int nn) -> float4 { return ltg_ld4(dpre1 + (size_t)min(k, kend - 1) * h12 + coff, nn, N, k < kend); };
        if constexpr (V == 3) ltg_gemm_block<MODE, TS, TS, (TS == 32 ? 128 : 64), 2, 2, true, true, false, FP8_S_EMB, FP8_S_G1, 3>(h0 + 1, N, m0, n0, kbeg, kend, a4, b4, epi);
        else ltg_gemm_block<MODE, TS, TS, 128, 2, 2, true, true, false, FP8_S_EMB, FP8_S_G1, 1>(h0 + 1, N, m0, n0, kbeg, kend, a4, b, epi);
    } else {
        ltg_gemm_block<MODE, TS, TS, 128, 2, 2, true, true, false, FP8_S_EMB, FP8_S_G1>(h0 + 1, N, m0, n0, kbeg, kend, a, b, epi);
    }
}

// One Adam sweep over all 8 discriminator tensors (train.py:163): g = sum of the split-K slabs.
// Block 0 additionally reduces the per-row loss terms into loss_out[0] (d_loss, train.py:142).
__global__ __launch_bounds__(NT) void k_d_adam(int ks, DLayout L, int SP, const float* __restrict__ slab, ltg_disc_state st, AdamC ad,
                                               int n, const float* __restrict__ lrow, float* __restrict__ loss_out) {
    __shared__ float red[NT / 64];
    const int P = L.off[8];     // SP = stride of a slab (>= P); lrow == nullptr: the loss sum sits in slot P of every slab
    for (int e = blockIdx.x * NT + threadIdx.x; e < P; e += gridDim.x * NT) {
        float g = 0.f;
        for (int z = 0; z < ks; ++z) g += slab[(size_t)z * SP + e];
        int t = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) t += e >= L.off[i] ? 1 : 0;
        const size_t i = (size_t)(e - L.off[t]);
        adam_update(st.p[t], st.m[t], st.v[t], i, g, ad);
        if (st.w1t_fp8 && (t == 0 || t == 2 || t == 4)) {   // operand-format shadow of the weight just written: [n][k], k contiguous
            const int nn_w = L.off[t + 2] - L.off[t + 1];   // row length of the weight = size of the bias that follows it
            const size_t k = i / nn_w, nn = i % nn_w;
            const size_t kdim = (size_t)(L.off[t + 1] - L.off[t]) / nn_w;
            uint8_t* dst = t == 0 ? st.w1t_fp8 : (t == 2 ? st.w2t_fp8 : st.w3t_fp8);
            dst[nn * kdim + k] = ltg_f2fp8(st.p[t][i] * (float)(1 << FP8_S_W));
            if (t == 4 && st.w3_fp8) st.w3_fp8[i] = ltg_f2fp8(st.p[t][i] * (float)(1 << FP8_S_W));   // w3 in its own layout (backward operand)
        }
    }
    if (blockIdx.x == 0) {
        float s = 0.f;
        if (lrow) for (int i = threadIdx.x; i < n; i += NT) s += lrow[i];
        else for (int z = threadIdx.x; z < ks; z += NT) s += slab[(size_t)z * SP + P];
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss_out[0] = s;
    }
}

// One gradient vector from the chunk slabs: out[e] = sum_z slab[z][e], out[P] = the loss sum (from lrow, or from slot P of the
// slabs) -- what a rank contributes to the gradient all-reduce when the pair rows are split over ranks (ltg_d_grad).
__global__ __launch_bounds__(NT) void k_d_grad_sum(int ks, int P, int stride, const float* __restrict__ slab, int n,
                                                   const float* __restrict__ lrow, float* __restrict__ out) {
    __shared__ float red[NT / 64];
    for (int e = blockIdx.x * NT + threadIdx.x; e < P; e += gridDim.x * NT) {
        float g = 0.f;
        for (int z = 0; z < ks; ++z) g += slab[(size_t)z * stride + e];
        out[e] = g;
    }
    if (blockIdx.x == 0) {
        float s = 0.f;
        if (lrow) for (int i = threadIdx.x; i < n; i += NT) s += lrow[i];
        else for (int z = threadIdx.x; z < ks; z += NT) s += slab[(size_t)z * stride + P];
        s = block_sum(s, red);
        if (threadIdx.x == 0) out[P] = s;
    }
}
