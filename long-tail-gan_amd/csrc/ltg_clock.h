// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): lazy Adam clock of W_q0.
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

// ---------------------------------------------------------------------------------------------
// Lazy Adam clock of W_q0 (ltg_gen_state.q0_last, include/ltg.h).  A zero-gradient Adam step of a row is
//     m <- b1 m,  v <- b2 v,  W <- W - lr_t m / (sqrt(v) + eps)
// -- the dense sweep's expressions with g == 0 (b1 m + (1-b1) 0 rounds once either way) -- so a row that lags k steps is
// brought up to date by running those k steps in registers: 24 B/parameter of traffic once per k steps instead of every
// step.  The arithmetic (k x IEEE sqrt and divide per parameter) does not shrink; it moves off the HBM stream.
// ---------------------------------------------------------------------------------------------
#define LTG_Q0_MASK (LTG_Q0_HIST - 1)

// One row (H4 float4 columns at W4/m4/v4) from ordinal `from` to ordinal `to`: zero-gradient steps, then -- if G4 is given --
// the step `to` itself with gradient row G4 and learning rate ad.lr_t (the caller's current step).
// (q0_row_steps: the zero-gradient steps from + 1 .. nz of one float4 column that is already in registers; true = it changed)
__device__ __forceinline__ bool q0_row_steps(float4& p, float4& mm, float4& vv, int from, int nz, const float* __restrict__ lr_hist, const AdamC ad) {
    const bool m0 = mm.x == 0.f && mm.y == 0.f && mm.z == 0.f && mm.w == 0.f;
    const bool v0 = vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
    if (m0 && v0) return false;                           // a row no batch has touched yet: every step is the identity
    if (m0) {                                             // W does not move (0 / (sqrt(v) + eps) == 0): only v decays
        for (int j = from + 1; j <= nz; ++j) { vv.x *= ad.b2; vv.y *= ad.b2; vv.z *= ad.b2; vv.w *= ad.b2; }
    } else {
        for (int j = from + 1; j <= nz; ++j) {
            const float lr = lr_hist[j & LTG_Q0_MASK];
#define LTG_ADAM0(f)          \
    mm.f = ad.b1 * mm.f;      \
    vv.f = ad.b2 * vv.f;      \
    p.f = adam_move(p.f, lr * mm.f, vv.f, ad.eps);   /* adam1 with g == 0: fma(b1, m, 0) rounds like b1 m */
            LTG_ADAM0(x) LTG_ADAM0(y) LTG_ADAM0(z) LTG_ADAM0(w)
#undef LTG_ADAM0
        }
    }
    return true;
}
// ONE zero-gradient step with the learning rate given (the step the caller is performing: its rate is not in the ring yet)
__device__ __forceinline__ bool q0_zero_step(float4& p, float4& mm, float4& vv, float lr, const AdamC ad) {
    const bool m0 = mm.x == 0.f && mm.y == 0.f && mm.z == 0.f && mm.w == 0.f;
    const bool v0 = vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
    if (m0 && v0) return false;
    if (m0) {
        vv.x *= ad.b2; vv.y *= ad.b2; vv.z *= ad.b2; vv.w *= ad.b2;
    } else {
#define LTG_ADAM0(f)          \
    mm.f = ad.b1 * mm.f;      \
    vv.f = ad.b2 * vv.f;      \
    p.f = adam_move(p.f, lr * mm.f, vv.f, ad.eps);
        LTG_ADAM0(x) LTG_ADAM0(y) LTG_ADAM0(z) LTG_ADAM0(w)
#undef LTG_ADAM0
    }
    return true;
}
__device__ __forceinline__ void q0_row_advance(float4* __restrict__ W4, float4* __restrict__ m4, float4* __restrict__ v4, int H4, int from, int to,
                                               const float* __restrict__ lr_hist, const float4* __restrict__ G4, const AdamC ad) {
    const int nz = G4 ? to - 1 : to;   // last zero-gradient step
    for (int c = threadIdx.x; c < H4; c += blockDim.x) {
        float4 p = W4[c], mm = m4[c], vv = v4[c];
        const bool moved = q0_row_steps(p, mm, vv, from, nz, lr_hist, ad);
        if (!G4 && !moved) continue;
        if (G4) {
            const float4 g = G4[c];
#define LTG_ADAM1(f) adam1(p.f, mm.f, vv.f, g.f, ad.lr_t, ad);
            LTG_ADAM1(x) LTG_ADAM1(y) LTG_ADAM1(z) LTG_ADAM1(w)
#undef LTG_ADAM1
        }
        W4[c] = p;
        m4[c] = mm;
        v4[c] = vv;
    }
}

#define Q0_NT 192
// rows of the batch's distinct items (G-step batches carry the list): up to `target`, before enc-0 reads them
__global__ __launch_bounds__(Q0_NT) void k_q0_touch_unique(int H, int nu, const int32_t* __restrict__ uptr, const int32_t* __restrict__ csr_pos,
                                                           const int32_t* __restrict__ indices, const int32_t* __restrict__ uitem, int target,
                                                           ltg_gen_state st, AdamC ad, const unsigned* __restrict__ poison = nullptr,
                                                           int32_t* __restrict__ mark = nullptr, unsigned seq = 0u) {
    // (one-call step, slice on the side stream: the slice of the previous call is done with every row before this kernel starts -- the
    // previous call's last kernel on this stream waited for word 6, ltg_gate_wait_tail; poison: that wait gave up)
    // mark: ltg_pipe.q0_mark -- "call seq's batch holds this row" for the ahead kernel of the same call (k_q0_touch_ahead)
    if (ltg_poisoned(poison)) return;
    const int u = blockIdx.x;
    if (u >= nu) return;
    const int H4 = H >> 2;
    if (uitem && H4 <= Q0_NT) {   // the item in one load; its clock and its row requested together (three dependent round trips, not six)
        const int i = uitem[u];
        const size_t off = (size_t)i * H4 + min((int)threadIdx.x, H4 - 1);
        const int from = st.q0_last[i];
        float4 p = reinterpret_cast<const float4*>(st.p[0])[off], mm = reinterpret_cast<const float4*>(st.m[0])[off],
               vv = reinterpret_cast<const float4*>(st.v[0])[off];
        if (mark && threadIdx.x == 0) mark[i] = (int32_t)seq;
        __syncthreads();   // every thread has read q0_last[i]
        if (from >= target) return;
        if ((int)threadIdx.x < H4 && q0_row_steps(p, mm, vv, from, target, st.q0_lr_hist, ad)) {
            reinterpret_cast<float4*>(st.p[0])[off] = p;
            reinterpret_cast<float4*>(st.m[0])[off] = mm;
            reinterpret_cast<float4*>(st.v[0])[off] = vv;
        }
        if (threadIdx.x == 0) st.q0_last[i] = target;
        return;
    }
    const int i = uitem ? uitem[u] : indices[csr_pos[uptr[u]]];
    const int from = st.q0_last[i];
    if (mark && threadIdx.x == 0) mark[i] = (int32_t)seq;
    if (from >= target) return;
    const size_t off = (size_t)i * H4;
    q0_row_advance(reinterpret_cast<float4*>(st.p[0]) + off, reinterpret_cast<float4*>(st.m[0]) + off, reinterpret_cast<float4*>(st.v[0]) + off, H4,
                   from, target, st.q0_lr_hist, nullptr, ad);
    __syncthreads();   // every thread has read q0_last[i]
    if (threadIdx.x == 0) st.q0_last[i] = target;
}

// The NEXT batch's rows, during the current call (ordinal `seq`, Adam step `cur` = q0_ord + 1), on the side stream behind the slice:
// up to `cur` -- zero-gradient steps from the ring up to cur - 1, then step cur itself with this call's learning rate (the sparse gradient
// kernel stores it into the ring, possibly later) -- for every row the CURRENT batch does not hold (mark != seq: nobody else reads or
// writes those rows during this call); the rows it holds reach `cur` through the sparse gradient kernel.  Either way the row is marked
// for the next call (seq + 1), whose catch-up launch the host then leaves out (ltg_pipe.caught_up).
__global__ __launch_bounds__(Q0_NT) void k_q0_touch_ahead(int H, int nu, const int32_t* __restrict__ uitem, int cur, ltg_gen_state st, AdamC ad,
                                                          int32_t* __restrict__ mark, unsigned seq, const unsigned* __restrict__ poison) {
    if (ltg_poisoned(poison)) return;
    const int u = blockIdx.x;
    if (u >= nu) return;
    const int H4 = H >> 2;
    const int i = uitem[u];
    const size_t off = (size_t)i * H4 + min((int)threadIdx.x, H4 - 1);
    const bool held = (unsigned)mark[i] == seq;
    const int from = st.q0_last[i];
    // (requested beside the mark and the clock; a held row's values may be mid-update by the sparse gradient kernel: they are discarded)
    float4 p = reinterpret_cast<const float4*>(st.p[0])[off], mm = reinterpret_cast<const float4*>(st.m[0])[off],
           vv = reinterpret_cast<const float4*>(st.v[0])[off];
    __syncthreads();   // every thread has read mark[i] and q0_last[i]
    if (threadIdx.x == 0) mark[i] = (int32_t)(seq + 1u);
    if (held || from >= cur) return;
    if ((int)threadIdx.x < H4) {
        bool moved = q0_row_steps(p, mm, vv, from, cur - 1, st.q0_lr_hist, ad);
        moved = q0_zero_step(p, mm, vv, ad.lr_t, ad) || moved;
        if (moved) {
            reinterpret_cast<float4*>(st.p[0])[off] = p;
            reinterpret_cast<float4*>(st.m[0])[off] = mm;
            reinterpret_cast<float4*>(st.v[0])[off] = vv;
        }
    }
    if (threadIdx.x == 0) st.q0_last[i] = cur;
}

// The catch-up of a batch's rows AND the rotating slice (rows start, start + stride, ...) in ONE launch, both up to `target`: a row
// that is in both sets belongs to the workgroup whose atomic max on its clock comes first (the other one sees `target` and leaves);
// the consumers are later launches.  The one-call step's form of the two kernels above and below (one launch, no side-stream join).
__global__ __launch_bounds__(Q0_NT) void k_q0_touch_slice(int I, int H, int nu, const int32_t* __restrict__ uptr, const int32_t* __restrict__ csr_pos,
                                                          const int32_t* __restrict__ indices, const int32_t* __restrict__ uitem, int target, int start,
                                                          int stride, ltg_gen_state st, AdamC ad) {
    __shared__ int s_from;
    const int b = blockIdx.x;
    int i;
    if (b < nu) i = uitem ? uitem[b] : indices[csr_pos[uptr[b]]];
    else {
        i = start + (b - nu) * stride;
        if (i >= I) return;
    }
    const int H4 = H >> 2;
    if (H4 <= Q0_NT) {   // the row requested beside the claim (rows are never written by two launches at once: whoever loses the claim
                         // only discards what it loaded)
        if (threadIdx.x == 0) s_from = atomicMax(st.q0_last + i, target);
        const size_t off = (size_t)i * H4 + min((int)threadIdx.x, H4 - 1);
        float4 p = reinterpret_cast<const float4*>(st.p[0])[off], mm = reinterpret_cast<const float4*>(st.m[0])[off],
               vv = reinterpret_cast<const float4*>(st.v[0])[off];
        __syncthreads();
        const int from = s_from;
        if (from >= target) return;
        if ((int)threadIdx.x < H4 && q0_row_steps(p, mm, vv, from, target, st.q0_lr_hist, ad)) {
            reinterpret_cast<float4*>(st.p[0])[off] = p;
            reinterpret_cast<float4*>(st.m[0])[off] = mm;
            reinterpret_cast<float4*>(st.v[0])[off] = vv;
        }
        return;
    }
    if (threadIdx.x == 0) s_from = atomicMax(st.q0_last + i, target);
    __syncthreads();
    const int from = s_from;
    if (from >= target) return;
    const size_t off = (size_t)i * H4;
    q0_row_advance(reinterpret_cast<float4*>(st.p[0]) + off, reinterpret_cast<float4*>(st.m[0]) + off, reinterpret_cast<float4*>(st.v[0]) + off, H4,
                   from, target, st.q0_lr_hist, nullptr, ad);
}

// forward-only batches (no distinct-item list): one workgroup per user row walks its entries; the first workgroup to claim
// a lagging item row (compare-and-swap on its clock) brings it up to date, the consumers are later launches
__global__ __launch_bounds__(Q0_NT) void k_q0_touch_rows(int H, int R, const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int target,
                                                         ltg_gen_state st, AdamC ad) {
    __shared__ int s_from;
    const int r = blockIdx.x;
    const int H4 = H >> 2;
    const int e1 = indptr[r + 1];
    for (int e = indptr[r]; e < e1; ++e) {
        const int i = indices[e];
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_load(st.q0_last + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_from = (old < target && atomicCAS(st.q0_last + i, old, target) == old) ? old : -1;
        }
        __syncthreads();
        const int from = s_from;
        __syncthreads();
        if (from < 0) continue;
        const size_t off = (size_t)i * H4;
        q0_row_advance(reinterpret_cast<float4*>(st.p[0]) + off, reinterpret_cast<float4*>(st.m[0]) + off, reinterpret_cast<float4*>(st.v[0]) + off,
                       H4, from, target, st.q0_lr_hist, nullptr, ad);
    }
}

// G step `ord` on the batch's rows: gradient row u of G (k_enc0_grad layout) for distinct item u; block nu = the bias row
// (dense: its gradient is never zero) + the learning rate of this step into the history ring
__global__ __launch_bounds__(Q0_NT) void k_q0_step_touched(int I, int H, int nu, const int32_t* __restrict__ uptr, const int32_t* __restrict__ csr_pos,
                                                           const int32_t* __restrict__ indices, const float* __restrict__ G, int ord, ltg_gen_state st,
                                                           AdamC ad) {
    const int u = blockIdx.x;
    const int H4 = H >> 2;
    const float4* G4 = reinterpret_cast<const float4*>(G);
    if (u == nu) {
        if (threadIdx.x == 0) st.q0_lr_hist[ord & LTG_Q0_MASK] = ad.lr_t;
        float4* b4 = reinterpret_cast<float4*>(st.p[4]);
        float4* mb4 = reinterpret_cast<float4*>(st.m[4]);
        float4* vb4 = reinterpret_cast<float4*>(st.v[4]);
        for (int c = threadIdx.x; c < H4; c += blockDim.x) {
            float4 g = G4[(size_t)nu * H4 + c];
#pragma unroll
            for (int j = 1; j < ENC0_BIAS_PARTS; ++j) {
                const float4 t = G4[(size_t)(nu + j) * H4 + c];
                g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
            }
            float4 p = b4[c], mm = mb4[c], vv = vb4[c];
#define LTG_ADAM1(f) adam1(p.f, mm.f, vv.f, g.f, ad.lr_t, ad);
            LTG_ADAM1(x) LTG_ADAM1(y) LTG_ADAM1(z) LTG_ADAM1(w)
#undef LTG_ADAM1
            b4[c] = p;
            mb4[c] = mm;
            vb4[c] = vv;
        }
        return;
    }
    const int i = indices[csr_pos[uptr[u]]];
    const int from = st.q0_last[i];
    const size_t off = (size_t)i * H4;
    q0_row_advance(reinterpret_cast<float4*>(st.p[0]) + off, reinterpret_cast<float4*>(st.m[0]) + off, reinterpret_cast<float4*>(st.v[0]) + off, H4,
                   from, ord, st.q0_lr_hist, G4 + (size_t)u * H4, ad);
    __syncthreads();
    if (threadIdx.x == 0) st.q0_last[i] = ord;
}

// rows start, start + stride, ...: zero-gradient steps up to `target` (the rotating slice of a G step; the flush: 0, 1)
__global__ __launch_bounds__(Q0_NT) void k_q0_sweep(int I, int H, int start, int stride, int target, ltg_gen_state st, AdamC ad,
                                                    const unsigned* __restrict__ poison = nullptr) {
    if (ltg_poisoned(poison)) return;
    const int H4 = H >> 2;
    for (size_t i = (size_t)start + (size_t)blockIdx.x * stride; i < (size_t)I; i += (size_t)gridDim.x * stride) {
        const int from = st.q0_last[i];
        __syncthreads();   // every thread has read the row's clock before thread 0 may move it
        if (from >= target) continue;
        const size_t off = i * H4;
        q0_row_advance(reinterpret_cast<float4*>(st.p[0]) + off, reinterpret_cast<float4*>(st.m[0]) + off, reinterpret_cast<float4*>(st.v[0]) + off,
                       H4, from, target, st.q0_lr_hist, nullptr, ad);
        if (threadIdx.x == 0) st.q0_last[i] = target;
    }
}
