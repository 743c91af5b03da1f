// HIP kernels + C ABI (include/ltg.h) of the MI355X-native Long-Tail-GAN training path.
// gfx950 only.  Reference citations are relative to /root/reference/.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/ltg.h"
#include "ltg_gemm.h"
#include "ltg_rng.h"

namespace {

constexpr int NT = 256;

struct AdamC {
    float lr_t, b1, b2, eps;
};

// tf.train.AdamOptimizer update (train.py:160-164, Q5): m,v updated for EVERY element (dense).
// One element, with the rounding PINNED (explicit fma; `b1 m + (1-b1) g` may otherwise contract either way, kernel by
// kernel): every kernel that applies Adam -- dense sweeps, tile epilogues, the lazy clock of W_q0 -- yields the same bits
// for the same (p, m, v, g).
// p - lr_t m / (sqrt(v) + eps).  Default: the hardware's square root and reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each, the
// quotient within ~2.5 ulp of the correctly rounded one -- 3e-7 of a step that is itself ~lr of the weight) instead of the IEEE
// sequences (~25 VALU instructions per element: what bounds the lazy clock's catch-up kernels and a fifth of the streaming weight
// update's issue slots).  The product with the reciprocal and the subtraction are ONE explicit fma: left to the compiler,
// `p - lm * r` contracts in one kernel and not in another.  -DLTG_ADAM_IEEE builds the correctly rounded form (A/B:
// scripts/build_variant.sh).  Every Adam update of the library goes through this one function, so dense sweep == lazy clock
// bit for bit either way.
__device__ __forceinline__ float adam_move(const float p, const float lm, const float v, const float eps) {
#ifdef LTG_ADAM_IEEE
    return p - lm / (sqrtf(v) + eps);
#else
    return __builtin_fmaf(-lm, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps), p);
#endif
}
__device__ __forceinline__ void adam1(float& p, float& m, float& v, const float g, const float lr_t, const AdamC& c) {
    m = __builtin_fmaf(c.b1, m, (1.f - c.b1) * g);
    v = __builtin_fmaf(c.b2, v, ((1.f - c.b2) * g) * g);
    p = adam_move(p, lr_t * m, v, c.eps);
}
__device__ __forceinline__ void adam_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                            size_t i, float g, const AdamC c) {
    float pn = p[i], mn = m[i], vn = v[i];
    adam1(pn, mn, vn, g, c.lr_t, c);
    m[i] = mn;
    v[i] = vn;
    p[i] = pn;
}

__device__ __forceinline__ float block_sum(float x, float* red /*[NT/64]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float x, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}

// ---- device-side hand-over between the two streams of the one-call step (include/ltg.h: ltg_pipe.sync).  A cross-stream event costs
// ~12 us per direction on this pool (scripts/micro/sync_cost.hip: hipEventRecord + hipStreamWaitEvent between two 10-us kernels) and
// ~6 us of bubble on the recording stream; a word in device memory costs the consumer one poll.  A gate is a 32-bit sequence number:
// the producer stores the call's ordinal, the consumer waits until the word has reached it (wrap-safe compare).  Every wait is bounded
// (30 s, counted in ltg_pipe.sync[2]): a call that failed half-way leaves a waiter behind, not a hung GPU.
// The two streams must be CONCURRENT: HIP maps streams onto a few hardware queues, and a waiter in front of its producer in one queue
// waits for ever -- ltg_g_pipe_probe tests a pair of streams for that.
struct LtgGate {
    unsigned* word;   // NULL: no gate
    unsigned seq;
    unsigned* expired;   // counts the waits that gave up (ltg_pipe.sync[2]: the host checks it when it joins the pipe)
    int limit;           // milliseconds before a wait gives up (0: 30 s)
};
#define LTG_NO_GATE LtgGate{nullptr, 0u, nullptr, 0}
// acquire = false: the consumer only needs to run AFTER the producer (a write-after-read hazard), it reads nothing the producer wrote --
// no cache invalidation (dec-0: 14.7 -> ~8 us; whoever reads the producer's data later does so behind a kernel boundary)
__device__ __forceinline__ bool ltg_poisoned_word(const unsigned* p) {
    return p && __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void ltg_gate_wait(LtgGate g, bool acquire = true) {   // first statement of a consumer kernel; every thread calls
    if (!g.word) return;
    if (threadIdx.x == 0) {
        bool open = false;
        const unsigned long long ticks = (unsigned long long)(g.limit > 0 ? g.limit : 30000) * 100000ull;   // wall_clock64: 100 MHz
        const unsigned long long t0 = wall_clock64();
        bool dead = false;     // the pipe is poisoned already: nothing behind this wait will touch the model, so nothing is waited for
        while (!open) {
            open = (int)(__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.seq) >= 0;
            dead = ltg_poisoned_word(g.expired);
            if (open || dead || wall_clock64() - t0 > ticks) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (!open && !dead && g.expired) atomicAdd(g.expired, 1u);
    }
    __syncthreads();
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// The same wait as the LAST thing ONE thread of a kernel does: the kernel behind it on the stream starts in stream order, i.e. after this
// one has ended, so that kernel is gated whatever its size while a single wave of the whole device spins (no workgroup of a large
// launch ever occupies a CU polling for a producer that still needs one: forward progress by construction).  A wait that gives up
// POISONS the pipe (word 2 != 0): the kernels behind it skip their work (ltg_poisoned) and the host raises when it next looks.
__device__ __forceinline__ void ltg_gate_wait_tail(LtgGate g) {
    if (!g.word) return;
    const unsigned long long ticks = (unsigned long long)(g.limit > 0 ? g.limit : 30000) * 100000ull;   // wall_clock64: 100 MHz
    const unsigned long long t0 = wall_clock64();
    bool open = false, dead = false;
    while (!open) {
        open = (int)(__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.seq) >= 0;
        dead = ltg_poisoned_word(g.expired);     // (poisoned already: the kernels behind this wait return at once -- nothing to wait for)
        if (open || dead || wall_clock64() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (!open && !dead && g.expired) atomicAdd(g.expired, 1u);
}
// word 2 of ltg_pipe.sync: a device-side wait of this pipe has given up -- nothing that follows may touch the model
// (round 5: a PLAIN load -- the scalar unit's, see ltg_poison_word below for why that is enough -- instead of an agent-scope atomic one: the
// guard is the first statement of its kernels, and as a vector load it was a ~1-us round trip in front of every other request)
// (round 6: the pointer is NOT __restrict__ -- waiters on other streams atomicAdd this word while the kernel runs, so a noalias promise would be
// false.  The guarantee is stated as what it is: a kernel sees every expiry that happened before it STARTED (its dispatch acquires); an expiry
// DURING the kernel is seen from the next kernel boundary on.  That is all the design needs: the wait that gates a kernel sits in front of it
// on its own stream.)
__device__ __forceinline__ bool ltg_poisoned(const unsigned* p) { return p && *p != 0u; }
// The word itself, for kernels that REQUEST it first and look at it in front of their first store (round 5): `if (ltg_poisoned(p)) return;`
// as a kernel's first statement is a vector-memory round trip of its own in front of every other request.  A PLAIN load of a uniform address
// before the kernel's first store: the compiler issues it on the SCALAR unit (s_load_dword, beside the kernel-argument loads), so nothing in
// the vector-memory queue waits for it.  Plain is enough here: every poisoner is a wait that sits IN FRONT of the reading kernel on its own
// stream (a kernel's end releases, a kernel's start acquires); a wait on another stream that gives up while this kernel runs is seen by the
// next kernel, as with the atomic load.
__device__ __forceinline__ unsigned ltg_poison_word(const unsigned* p) { return p ? *p : 0u; }
__device__ __forceinline__ bool ltg_word_set(unsigned w) { return w != 0u; }
// by ONE thread.  Every producer in this library is a WHOLE KERNEL that ended in front of the kernel that stores the word (the store
// is the first thing a kernel does when it starts, or a one-wave kernel of its own behind the producer), and the end of a kernel is
// already the device-wide release of what it wrote; the agent-scope release below is belt and braces -- measured in round 4 against a
// build without it: bit-identical over 3 400 soak steps and no timing difference (profiles/r4_ab_gate_fence.txt), so it stays.
__device__ __forceinline__ void ltg_gate_set(LtgGate g) {
    if (!g.word) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(g.word, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave in front of the side stream's work: returns when the gate opens (the kernels behind it start in stream order)
// (set_first: a gate this kernel opens when it starts -- whatever preceded it on its stream is complete)
__global__ __launch_bounds__(64) void k_gate_wait(LtgGate g, LtgGate set_first = LTG_NO_GATE) {
    if (threadIdx.x == 0) ltg_gate_set(set_first);
    ltg_gate_wait(g);
}
// one wave behind the side stream's work: opens the gate
__global__ __launch_bounds__(64) void k_gate_set(LtgGate g, LtgGate g2 = LTG_NO_GATE) {
    if (threadIdx.x == 0) {
        ltg_gate_set(g);
        ltg_gate_set(g2);
    }
}


// ---------------------------------------------------------------------------------------------
// Generator forward
// ---------------------------------------------------------------------------------------------

// enc-0 as a sparse row gather-sum (MultiVAE.py:148-155): h1 = tanh(dropout(l2norm(x)) . W_q0 + b).
// One 1024-thread workgroup per user row.  The row's (item, value*keep) list is staged through LDS;
// the 16 waves split the row's entries (so a 900-item history does not serialise on one wave), each
// lane owning float4 column chunks of the gathered W_q0 rows (coalesced 16-B loads); the wave
// partials meet in LDS.
constexpr int ENC_NT = 1024;
constexpr int ENC_NW = ENC_NT / 64;
__global__ __launch_bounds__(ENC_NT) void k_enc0_fwd(int H, int I, const int32_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                     const uint8_t* __restrict__ drop_keep, float keep, uint64_t seed,
                                                     uint64_t step, const float* __restrict__ Wq0,
                                                     const float* __restrict__ bq0, float* __restrict__ h1,
                                                     float* __restrict__ row_scale, const float* __restrict__ row_norm2,
                                                     int item_lo, int Ig, int pre_only, int rps) {
    // item shard: `indices` are LOCAL item ids of this rank's slab [item_lo, item_lo + I); the dropout
    // RNG is keyed by the GLOBAL id so every shard draws the mask the unsharded run draws; row_norm2
    // (sum x^2 over the FULL row) replaces the local sum; pre_only writes the partial pre-activation
    // (no bias, no tanh) that the ranks all-reduce.
    extern __shared__ __attribute__((aligned(16))) float s_part[];  // [ENC_NW][H]
    __shared__ int s_idx[ENC_NT];
    __shared__ float s_val[ENC_NT];
    __shared__ float red[ENC_NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t kb = rps > 0 ? (uint64_t)(b % rps) : (uint64_t)b;   // several batches in one launch (ltg_fwd_opts.rows_per_step)
    step += rps > 0 ? (uint64_t)(b / rps) : 0;
    const int beg = indptr[b], end = indptr[b + 1];
    float ss = 0.f;
    for (int e = beg + tid; e < end; e += ENC_NT) {
        const float v = values ? values[e] : 1.f;
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red[w] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < ENC_NW; ++i) ss += red[i];
    if (row_norm2) ss = row_norm2[b];
    const float scale = 1.f / (keep * sqrtf(fmaxf(ss, 1e-12f)));  // l2_normalize eps, then /keep
    if (tid == 0) row_scale[b] = scale;
    const int H4 = H >> 2;
    constexpr int MAXQ = 4;  // H <= 1024
    float4 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = beg; c0 < end; c0 += ENC_NT) {
        __syncthreads();
        const int e = c0 + tid;
        if (e < end) {
            const int it = indices[e];
            const float v = values ? values[e] : 1.f;
            const bool kp = drop_keep ? (drop_keep[e] != 0)
                                      : ltg_rng_keep(seed, LTG_STREAM_VAE_DROPOUT, step, kb * (uint64_t)Ig + item_lo + it, keep);
            s_idx[tid] = it;
            s_val[tid] = kp ? v : 0.f;
        }
        __syncthreads();
        const int cnt = min(ENC_NT, end - c0);
        // 4 entries per trip: their W_q0 row loads are independent, so 4 x MAXQ float4 loads are in flight
        for (int j = w; j < cnt; j += 4 * ENC_NW) {
            float v[4];
            const float4* wr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ju = j + u * ENC_NW;
                const bool ok = ju < cnt;
                v[u] = ok ? s_val[ju] : 0.f;
                wr[u] = reinterpret_cast<const float4*>(Wq0 + (size_t)s_idx[ok ? ju : j] * H);
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int c4 = lane + 64 * q;
                if (c4 < H4) {
                    float4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = wr[u][c4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[q].x += v[u] * x[u].x;
                        acc[q].y += v[u] * x[u].y;
                        acc[q].z += v[u] * x[u].z;
                        acc[q].w += v[u] * x[u].w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < H4) reinterpret_cast<float4*>(s_part + (size_t)w * H)[c4] = acc[q];
    }
    __syncthreads();
    for (int c = tid; c < H; c += ENC_NT) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < ENC_NW; ++i) t += s_part[(size_t)i * H + c];
        h1[(size_t)b * H + c] = pre_only ? t * scale : tanhf(t * scale + bq0[c]);
    }
}

// h1 = tanh(h1_pre + b) after the partial pre-activations of the item shards were all-reduced
__global__ __launch_bounds__(NT) void k_bias_tanh(int n, int H, const float* __restrict__ bias, float* __restrict__ h) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) h[i] = tanhf(h[i] + bias[i % H]);
}

// float4 of 4 consecutive floats at p[i .. i+3], zero where i + j >= n (n % 4 == 0 at every call site, so a group is
// either whole or absent; the address is clamped, the load unconditional)
__device__ __forceinline__ float4 ltg_ld4(const float* __restrict__ p, int i, int n, bool ok) {
    const float4 v = *reinterpret_cast<const float4*>(p + min(i, n - 4));
    const bool k = ok && i < n;
    return make_float4(k ? v.x : 0.f, k ? v.y : 0.f, k ? v.z : 0.f, k ? v.w : 0.f);
}

// Generic dense layer  C = act(A[M][K] . B[K][N] + bias)  (fp32 MFMA); act: 0 none, 1 tanh.
// Serves enc-1 (MultiVAE.py:152) and dec-0 (MultiVAE.py:168-172).
template <int ACT, bool V, int BKV = 128>
__global__ __launch_bounds__(NT) void k_dense_fwd(int M, int N, int K, const float* __restrict__ A,
                                                  const float* __restrict__ Bw, const float* __restrict__ bias,
                                                  float* __restrict__ C) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    auto a = [=] __device__(int m, int k) -> float { return A[(size_t)m * K + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Bw[(size_t)k * N + n]; };
    auto epi = [=] __device__(int m, int n, float acc) {
        const float x = acc + bias[n];
        C[(size_t)m * N + n] = ACT == 1 ? tanhf(x) : x;
    };
    if constexpr (V) {   // 16-B loads (K % 4 == 0, N % 4 == 0): same MFMA sequence, a quarter of the load instructions
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(A + (size_t)min(m, M - 1) * K, k, K, m < M); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Bw + (size_t)min(k, K - 1) * N, n, N, k < K); };
        ltg_gemm_block<false, 32, 32, BKV, 2, 2, false, true, false, 0, 0, 3>(M, N, m0, n0, 0, K, a4, b4, epi);
    } else {
        ltg_gemm_block<false, 32, 32, 128, 2, 2, false, true>(M, N, m0, n0, 0, K, a, b, epi);
    }
}

// Reparameterisation + KL (MultiVAE.py:157-162, :178-181).
__global__ __launch_bounds__(NT) void k_reparam(int Z, const float* __restrict__ mulv, const float* __restrict__ eps_in,
                                                float is_training, uint64_t seed, uint64_t step,
                                                float* __restrict__ z, float* __restrict__ kl_rows) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    float kl = 0.f;
    for (int j = threadIdx.x; j < Z; j += NT) {
        const float mu = mulv[(size_t)b * 2 * Z + j], lv = mulv[(size_t)b * 2 * Z + Z + j];
        const float sd = expf(0.5f * lv);
        kl += 0.5f * (-lv + expf(lv) + mu * mu - 1.f);
        float e = 0.f;
        if (is_training != 0.f)
            e = eps_in ? eps_in[(size_t)b * Z + j] : ltg_rng_normal(seed, LTG_STREAM_VAE_EPS, step, (uint64_t)b * Z + j);
        z[(size_t)b * Z + j] = mu + is_training * e * sd;
    }
    kl = block_sum(kl, red);
    if (threadIdx.x == 0) kl_rows[b] = kl;
}

// dec-1 (MultiVAE.py:169): logits[b][i] = h2[b][:] . W_p1t[i][:] + b_p1[i]; the big GEMM.
template <bool BF16, bool BIG, bool V = false>
__global__ __launch_bounds__(NT) void k_dec1_fwd(int M, int I, int H, const float* __restrict__ h2,
                                                 const float* __restrict__ Wp1t, const float* __restrict__ bp1,
                                                 float* __restrict__ logits) {
    // BIG: 128 x 64 tiles (a whole training batch per tile: W_p1t leaves HBM once); small item counts
    // use 32 x 32 tiles to spread the few tiles over more CUs.
    constexpr int BM = BIG ? 128 : 32, BN = BIG ? 64 : 32;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    auto a = [=] __device__(int m, int k) -> float { return h2[(size_t)m * H + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Wp1t[(size_t)n * H + k]; };
    auto epi = [=] __device__(int m, int n, float acc) { logits[(size_t)m * I + n] = acc + bp1[n]; };
    if constexpr (V) {   // 16-B loaders (H % 4 == 0)
        auto a4 = [=] __device__(int m, int k) -> float4 { return ltg_ld4(h2 + (size_t)min(m, M - 1) * H, k, H, m < M); };
        auto b4 = [=] __device__(int k, int n) -> float4 { return ltg_ld4(Wp1t + (size_t)min(n, I - 1) * H, k, H, n < I); };
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, false, false, 0, 0, 3>(M, I, m0, n0, 0, H, a4, b4, epi);
    } else {
        ltg_gemm_block<BF16, BM, BN, (BIG ? 64 : 128), 2, 2, false, false>(M, I, m0, n0, 0, H, a, b, epi);
    }
}


// ---------------------------------------------------------------------------------------------
// Streaming decoder kernels for large item counts (training batch <= 128 rows, H <= 608, bf16 MFMA).
// Both read W_p1t exactly once from HBM with 16-B loads, convert to bf16 on the fly into a double-
// buffered LDS tile of 32 items, and keep the small operand stationary in registers:
//   k_dec1_fwd_stream : h2 fragments stationary (each of the 8 waves owns 16 batch rows),
//                       logits[b][i] = h2[b][:] . W_p1t[i][:] + b_p1[i]
//   k_dh2_stream      : the [16 rows x 608] accumulator of each wave stationary over its item chunk,
//                       dh2[b][:] += dlog[b][i] * W_p1t[i][:]; W is consumed TRANSPOSED straight from its
//                       row-major LDS image by ds_read_b64_tr_b16 (no transposing stores)
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short ltg_s16x4;
constexpr int ST_NT = 512;      // 8 waves
constexpr int ST_BN = 32;       // items per LDS tile
constexpr int ST_KP = 608;      // K padded to 19 x 32
constexpr int ST_LDW = 616;     // LDS row stride in bf16 (1232 B: 16-B aligned rows, conflict-free fragment reads)
constexpr int ST_KS = ST_KP / 32;

// A wave-uniform GLOBAL pointer pinned to SGPRs: `ltg_uniform_ptr(base + uniform) + (unsigned)lane_offset` selects the
// scalar-base form of global_load/store (one 32-bit VGPR offset) instead of a 64-bit VGPR address per access.
typedef char __attribute__((address_space(1))) ltg_gchar;
typedef unsigned ltg_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ltg_gchar* ltg_uniform_ptr(const void* p) {
    const uint64_t x = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return (ltg_gchar*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint2 ltg_pack4(float4 v) {
    return make_uint2((unsigned)ltg_f2bf(v.x) | ((unsigned)ltg_f2bf(v.y) << 16), (unsigned)ltg_f2bf(v.z) | ((unsigned)ltg_f2bf(v.w) << 16));
}

// The streaming kernels read the bf16 SHADOW of W_p1t ([I][ST_KP] bf16, K padded with zeros, maintained by the
// Adam epilogue of k_dec1_bwd_adam): 1216 B per item instead of 2400, no conversion on the hot path.
// global -> registers for one 32-item tile: 16 threads walk one item row in 256-B steps, 5 x 16 B per thread.
constexpr int ST_C16 = ST_KP * 2 / 16;  // 76 16-byte chunks per shadow row
// (five named members, not an array: a conditionally written register array is demoted to scratch)
typedef __attribute__((ext_vector_type(4))) unsigned int ltg_u32x4;  // native vector: stays in VGPRs (HIP's uint4 is a union struct)
struct StW {
    ltg_u32x4 a, b, c, d, e;
};
__device__ __forceinline__ void st_fetch_w(const unsigned short* __restrict__ Wb, int I, int i0, StW& r) {
    const int it = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    const ltg_u32x4* row = reinterpret_cast<const ltg_u32x4*>(Wb + (size_t)min(i0 + it, I - 1) * ST_KP);
    r.a = row[c0];
    r.b = row[c0 + 16];
    r.c = row[c0 + 32];
    r.d = row[c0 + 48];
    r.e = row[min(c0 + 64, ST_C16 - 1)];
}
// one HALF of the shadow rows (the dh2 product split over column halves): 38 chunks of 16 B per item row, stored compactly (LDS columns
// 0 .. 303); lanes 6 .. 15 of a row's 16 threads have no third chunk (clamped duplicate load, no store)
__device__ __forceinline__ void st_fetch_w_half(const unsigned short* __restrict__ Wb, int I, int i0, int half, StW& r) {
    const int it = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    const ltg_u32x4* row = reinterpret_cast<const ltg_u32x4*>(Wb + (size_t)min(i0 + it, I - 1) * ST_KP) + half * (ST_C16 / 2);
    r.a = row[c0];
    r.b = row[c0 + 16];
    r.c = row[min(c0 + 32, ST_C16 / 2 - 1)];
}
__device__ __forceinline__ void st_stash_w_half(unsigned short* __restrict__ Wl, const StW& r) {
    const int it = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    ltg_u32x4* row = reinterpret_cast<ltg_u32x4*>(Wl + it * ST_LDW);
    row[c0] = r.a;
    row[c0 + 16] = r.b;
    if (c0 + 32 < ST_C16 / 2) row[c0 + 32] = r.c;
}
__device__ __forceinline__ void st_stash_w(unsigned short* __restrict__ Wl, const StW& r) {
    const int it = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    ltg_u32x4* row = reinterpret_cast<ltg_u32x4*>(Wl + it * ST_LDW);
    row[c0] = r.a;
    row[c0 + 16] = r.b;
    row[c0 + 32] = r.c;
    row[c0 + 48] = r.d;
    if (c0 + 64 < ST_C16) row[c0 + 64] = r.e;
}

// STATS: every lane also keeps the running (max, sum of exp) of the logits it stores (four batch rows x two items per tile);
// at the end the 16 lanes of a row merge theirs and the workgroup writes stat[blockIdx.x][row] = (max, sum exp(. - max)) over
// ITS tiles -- the softmax statistics come out of the producing epilogue, the [B, I] logits are not read again for them
// (k_row_stats_merge folds the workgroups' pairs and adds the sparse terms).
// PF: W tiles of HBM loads in flight per workgroup (register sets of 20 VGPRs each).  Measured at 200 000 items (85 us): PF = 3
// changes nothing; without the logits stores 69 us, with 1 of the 19 MFMA / LDS-read rounds 66 us, with neither 56 us (the
// 243 MB of W at 4.3 TB/s): loads, product and stores add up rather than overlap -- one lock-step workgroup per CU.
template <bool STATS, int PF = 2>
__global__ __launch_bounds__(ST_NT) void k_dec1_fwd_stream(int M, int I, int H, const float* __restrict__ h2,
                                                           const unsigned short* __restrict__ Wb, const float* __restrict__ bp1,
                                                           float* __restrict__ logits, float* __restrict__ stat) {
    extern __shared__ __attribute__((aligned(16))) unsigned short st_lds[];  // 2 x [ST_BN][ST_LDW]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    // stationary A fragments: rows 16w + lr, all K (fp32 -> bf16 once per workgroup).  Rows >= M MIRROR row M - 1:
    // their products equal row M - 1's and are stored to row M - 1's addresses (same value twice) -- no row predicate
    // anywhere in the loop, so every s_waitcnt is an exact count and the prefetch is never drained.
    // (round 5: the first two / three W tiles are requested BEFORE the h2 fragments, the first one goes to LDS behind them -- h2, then the first
    // tile, then the others was three dependent round trips in front of the first product of workgroups that own two or three tiles in all)
    const int ntiles = (I + ST_BN - 1) / ST_BN, G = gridDim.x, last = ntiles - 1;
    StW r0, r1, r2;
    int t = blockIdx.x, cur = 0;
    if (t < ntiles) {
        st_fetch_w(Wb, I, t * ST_BN, r0);
        st_fetch_w(Wb, I, min(t + G, last) * ST_BN, r1);
        if constexpr (PF == 3) st_fetch_w(Wb, I, min(t + 2 * G, last) * ST_BN, r2);
    }
    ltg_bf16x8 af[ST_KS];
    {
        // (the fragments in TWO batches of requests: all 38 at once beside the W tiles took 248 registers, and two such waves per SIMD leave the
        // side stream's clock kernels no room -- see DESIGN 5.3)
        const int row = min(16 * w + lr, M - 1);
        const float4* hr = reinterpret_cast<const float4*>(h2 + (size_t)row * H);
        const int H4 = H >> 2;
        constexpr int KSA = (ST_KS + 1) / 2;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float4 x0[KSA], x1[KSA];
#pragma unroll
            for (int j = 0; j < KSA; ++j) {
                const int ks = hb * KSA + j;
                if (ks < ST_KS) {
                    const int c4 = ks * 8 + 2 * lq;
                    x0[j] = hr[min(c4, H4 - 1)];
                    x1[j] = hr[min(c4 + 1, H4 - 1)];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < KSA; ++j) {
                const int ks = hb * KSA + j;
                if (ks < ST_KS) {
                    const int c4 = ks * 8 + 2 * lq;
                    const uint2 p0 = ltg_pack4(x0[j]), p1 = ltg_pack4(x1[j]);
                    const unsigned k0 = c4 < H4 ? 0xFFFFFFFFu : 0u, k1 = c4 + 1 < H4 ? 0xFFFFFFFFu : 0u;   // K padding -> 0
                    ltg_u32x4 tt;
                    tt[0] = p0.x & k0; tt[1] = p0.y & k0; tt[2] = p1.x & k1; tt[3] = p1.y & k1;
                    if (hb == 0) asm volatile("" : "+v"(tt[0]), "+v"(tt[1]), "+v"(tt[2]), "+v"(tt[3]));   // (packed HERE: the compiler otherwise sinks the packing behind the second batch)
                    af[ks] = __builtin_bit_cast(ltg_bf16x8, tt);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float rm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < ntiles) st_stash_w(st_lds, r0);
    __syncthreads();
    // ST_STEP(RL, RS): LDS[cur] holds tile tc = min(t, last), RS holds tile min(t + G, last) (in flight since the previous
    // step); tile min(t + 2G, last) is requested into RL, so two tiles of HBM loads are always outstanding per workgroup.
    // Tile indices are clamped instead of guarded: a step past the end recomputes the last tile and stores the same
    // logits again.  (A macro, not a lambda: register arrays captured by reference end up in scratch.)
#define ST_STEP(RL, RS)                                                                                                         \
    {                                                                                                                           \
        const int tc = min(t, last);                                                                                            \
        const int ia = min(tc * ST_BN + lr, I - 1), ib = min(tc * ST_BN + 16 + lr, I - 1);                                      \
        const float biasa = bp1[ia], biasb = bp1[ib]; /* BEFORE the prefetch: waiting for a younger load drains it */           \
        st_fetch_w(Wb, I, min(t + PF * G, last) * ST_BN, RL);                                                                   \
        const unsigned short* Wl = st_lds + cur * ST_BN * ST_LDW;                                                               \
        ltg_f32x4 acc0 = ltg_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = ltg_f32x4{0.f, 0.f, 0.f, 0.f};                                   \
        _Pragma("unroll") for (int ks = 0; ks < ST_KS; ++ks) {                                                                  \
            const ltg_u16x8 b0 = *reinterpret_cast<const ltg_u16x8*>(Wl + lr * ST_LDW + ks * 32 + 8 * lq);                      \
            const ltg_u16x8 b1 = *reinterpret_cast<const ltg_u16x8*>(Wl + (16 + lr) * ST_LDW + ks * 32 + 8 * lq);               \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], __builtin_bit_cast(ltg_bf16x8, b0), acc0, 0, 0, 0);          \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], __builtin_bit_cast(ltg_bf16x8, b1), acc1, 0, 0, 0);          \
        }                                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                         \
            const size_t ro = (size_t)min(16 * w + 4 * lq + q, M - 1) * I;                                                      \
            const float la = acc0[q] + biasa, lb = acc1[q] + biasb;                                                             \
            logits[ro + ia] = la;                                                                                               \
            logits[ro + ib] = lb;                                                                                               \
            if constexpr (STATS) { /* a clamped repeat of the last tile / an item past the end counts for nothing */            \
                const float xa = (t <= last && tc * ST_BN + lr < I) ? la : -INFINITY;                                           \
                const float xb = (t <= last && tc * ST_BN + 16 + lr < I) ? lb : -INFINITY;                                      \
                const float mn = fmaxf(rm[q], fmaxf(xa, xb)), mr = fmaxf(mn, -1e30f);                                           \
                rs[q] = rs[q] * __expf(rm[q] - mr) + __expf(xa - mr) + __expf(xb - mr);                                         \
                rm[q] = mn;                                                                                                     \
            }                                                                                                                   \
        }                                                                                                                       \
        st_stash_w(st_lds + (cur ^ 1) * ST_BN * ST_LDW, RS);                                                                    \
        __syncthreads();                                                                                                        \
        cur ^= 1;                                                                                                               \
    }
    if constexpr (PF == 3) {
        for (; t < ntiles; t += 3 * G) {
            ST_STEP(r0, r1)
            t += G;
            ST_STEP(r1, r2)
            t += G;
            ST_STEP(r2, r0)
            t -= 2 * G;
        }
    } else {
        for (; t < ntiles; t += 2 * G) {
            ST_STEP(r0, r1)
            t += G;
            ST_STEP(r1, r0)
            t -= G;
        }
    }
#undef ST_STEP
    if constexpr (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {   // the 16 lanes lr of a row
                const float m2 = __shfl_xor(rm[q], o), s2 = __shfl_xor(rs[q], o);
                const float mn = fmaxf(rm[q], m2), mr = fmaxf(mn, -1e30f);
                rs[q] = rs[q] * __expf(rm[q] - mr) + s2 * __expf(m2 - mr);
                rm[q] = mn;
            }
            const int row = 16 * w + 4 * lq + q;
            if (lr == 0 && row < M) {
                float* o2 = stat + ((size_t)blockIdx.x * M + row) * 2;
                o2[0] = rm[q];
                o2[1] = rs[q];
            }
        }
    }
}

// ---- The streaming decoder forward, second form (round 5; slabs of 65 536 items or more): WHO OWNS WHAT is turned round.  Above, the
// eight waves of a workgroup split the BATCH rows, so every wave needs every W tile: the tile goes through LDS, each wave reads all 39 KB
// of it (311 KB of LDS reads per 32 items and CU), and one barrier per tile keeps the eight waves in lock step -- loads, product and
// stores add up (56 + 13 + 16 us at 200 000 items).  Here h2 -- the SMALL operand, 100 x 608 bf16 -- is resident in LDS for the whole
// kernel, laid out in fragment order (every fragment read is one contiguous, conflict-free 1-KiB ds_read_b128), and each WAVE owns its
// own 32-item tiles:
//   * items are the M dimension of v_mfma_f32_16x16x32_bf16 (A = W_p1t shadow rows, B = h2^T): a lane's A fragment is 16 contiguous
//     bytes of ONE shadow row, loaded global -> VGPR in fragment order (16 rows x 64 B per wave instruction: every byte of a 128-B
//     line is used by two consecutive K steps) -- no LDS staging of W, no barrier in the loop, the waves drift apart and one wave's
//     loads overlap another's MFMAs and stores;
//   * two 16-item sub-tiles per wave tile share every B fragment read: 133 KB of LDS reads per 32 items and wave (NTB = 7) instead of
//     311 KB per 32 items and workgroup -- 2.3x fewer LDS bytes per item;
//   * W travels through a RING of 19 load units (one unit = one wave instruction = 16 B per lane = one (K step, sub-tile) fragment;
//     a tile is 38 units): the unit consumed by K step ks is re-requested for 19 units ahead -- the same tile's second half, then the
//     NEXT tile's first half -- so 19 KB per wave = 152 KB per CU of HBM loads are in flight at every moment, through the epilogue's
//     stores and across tile boundaries, with no register-set swap (19 is odd: unit u and u + 19 sit in the same registers);
//   * the accumulator of a lane is four CONSECUTIVE items of one batch row: logits leave as 16-B stores, 64 B contiguous per row and
//     instruction (128 B per row over the two sub-tiles), and the softmax statistics stay per lane (one (max, sum exp) pair per batch
//     tile), merged over the four lane groups and the eight waves once at the end -- same stat[workgroup][row] = (max, sum) output;
//   * the bias of a tile goes through the SCALAR unit (one s_load of the tile's 32 values, the lane picks its two groups of four with
//     bit masks): as a vector load it joins the in-order vmcnt queue wherever the compiler sinks it -- K step 14 -- and the wait for it
//     in front of the stores then drains 15 of the ring's 19 units.
// No branch in the loop: indices are clamped (a wave's last tile re-requests the tile it has just read instead of a next one: L2 hits),
// rows >= M mirror row M - 1, items >= I mirror the slab's last four, so every s_waitcnt is an exact count.
// Where it is used, and why not everywhere (round 5, profiles/README.md): ALONE on the chip it runs 74.7 us at 200 000 items against the
// first form's 79.2 (4.33 TB/s of a box that copies at 5.2) and 17.9 against 19.9 us at 25 024; INSIDE the one-call step of a 20 000- /
// 25 024-item slab the step got 1-3 / 4 us LONGER with it: its two waves per SIMD take all 512 registers (256 each), so the side
// stream's clock kernels (26-30 VGPRs), which the first form (2 x 224) leaves room for, wait for whole CUs to drain.  A 16-item-tile
// variant held to 168 VGPRs co-resides again but reads twice the LDS bytes per item: 85 us alone at 200 000 items, no gain in any step.
// So: this form for the HBM-bound slabs, the first form below 65 536 items.
// NTB: 16-row batch tiles (7 for <= 112 rows: the 100-row batches of config.ini; 8 up to 128 rows).
constexpr int ST2_UNITS = 2 * ST_KS;   // 38 load units per 32-item tile
constexpr int ST2_RING = ST_KS;        // 19 units in flight per wave
constexpr int ST2_MIN_ITEMS = 65536;
template <bool STATS, int NTB>
__global__ __launch_bounds__(ST_NT) void k_dec1_fwd_stream2(int M, int I, int H, const float* __restrict__ h2,
                                                            const unsigned short* __restrict__ Wb, const float* __restrict__ bp1,
                                                            float* __restrict__ logits, float* __restrict__ stat) {
    extern __shared__ __attribute__((aligned(16))) ltg_u32x4 Hs[];   // h2 in B-fragment order: [K step][batch tile][lane] x 16 B = 19 NTB KiB
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int ntiles = (I + 31) >> 5, G = gridDim.x, GW = 8 * G;
    // tiles are dealt to waves CU-first (wave w of workgroup g is wave number w * G + g): a slab with fewer tiles than waves spreads over
    // all CUs, a few waves each, instead of filling some CUs with eight waves
    int t = w * G + (int)blockIdx.x;
    const bool any = t < ntiles;
    const ltg_gchar* Wg = ltg_uniform_ptr(Wb);
    ltg_gchar* Lg = ltg_uniform_ptr(logits);      // (32-bit byte offsets into BOTH: the host sends launches of 2^30 logits or more, or of slabs whose shadow
                                                  // is 2^32 bytes or more -- I >= 3 532 111 --, to the first form: launch_dec1_fwd_stream)
    // byte offset of this lane's A-fragment row in the shadow: item row (tile, sub-tile s, lr), chunk lq of the K step (the K step's
    // 64 B are added as an immediate).  Items past the end mirror the slab's last four (I % 4 == 0: a lane's four output items are all
    // inside or all outside): the product of a mirrored row group is the last group's, and it is stored to the last group's address.
    auto rowoff = [&](int tt, int ss) -> unsigned {
        int it = tt * 32 + 16 * ss + lr;
        it = it < I ? it : I - 4 + (it & 3);
        return (unsigned)it * (unsigned)(ST_KP * 2) + 16u * (unsigned)lq;
    };
    ltg_u32x4 Wr[ST2_RING];
    typedef const ltg_u32x4 __attribute__((address_space(1))) * st2_gp;
#define ST2_LOAD(u, OFF0, OFF1) Wr[(u) % ST2_RING] = *(st2_gp)(Wg + (((u) & 1) ? (OFF1) : (OFF0)) + 64u * (unsigned)((u) >> 1));
    unsigned c0 = any ? rowoff(t, 0) : 0u, c1 = any ? rowoff(t, 1) : 0u;
    // the ring's first 19 units BEFORE the prologue: the first HBM round trip runs under the construction of the h2 image
#pragma unroll
    for (int u = 0; u < ST2_RING; ++u) { ST2_LOAD(u, c0, c1) }
    {   // h2 (fp32, [M][H]) -> bf16 fragments in LDS; rows >= M mirror row M - 1, columns >= H are zero (K padding)
        const int H4 = H >> 2;
        constexpr int NE = ST_KS * NTB * 64, PER = (NE + ST_NT - 1) / ST_NT, CH = 6;
#pragma unroll 1
        for (int j0 = 0; j0 < PER; j0 += CH) {
            float4 x0[CH], x1[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int e = min(tid + (j0 + j) * ST_NT, NE - 1), ln = e & 63, fr = e >> 6, nt = fr % NTB, ks = fr / NTB;
                const int row = min(16 * nt + (ln & 15), M - 1), c4 = ks * 8 + 2 * (ln >> 4);
                const float4* hr = reinterpret_cast<const float4*>(h2 + (size_t)row * H);
                x0[j] = hr[min(c4, H4 - 1)];
                x1[j] = hr[min(c4 + 1, H4 - 1)];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int e = tid + (j0 + j) * ST_NT, fr = min(e, NE - 1) >> 6, ks = fr / NTB, c4 = ks * 8 + 2 * ((e & 63) >> 4);
                const uint2 p0 = ltg_pack4(x0[j]), p1 = ltg_pack4(x1[j]);
                const unsigned k0 = c4 < H4 ? 0xFFFFFFFFu : 0u, k1 = c4 + 1 < H4 ? 0xFFFFFFFFu : 0u;
                ltg_u32x4 v;
                v[0] = p0.x & k0; v[1] = p0.y & k0; v[2] = p1.x & k1; v[3] = p1.y & k1;
                if (e < NE && j0 + j < PER) Hs[e] = v;
            }
        }
    }
    __syncthreads();
    float rm[NTB], rs[NTB];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
        rm[nt] = -INFINITY;
        rs[nt] = 0.f;
    }
    if (any) {
        const ltg_u32x4* Hl = Hs + lane;
#pragma unroll 1
        for (; t < ntiles; t += GW) {
            const bool more = t + GW < ntiles;
            // the next tile's row offsets; on the wave's last tile the ring re-requests the tile it has just read (a uniform select, no branch:
            // a join of divergent paths would cost every wait its exact count): served by the L2, no HBM traffic
            const int tnx = more ? t + GW : t;
            const unsigned n0 = rowoff(tnx, 0), n1 = rowoff(tnx, 1);
            const int i0 = t * 32 + 4 * lq, i1 = i0 + 16;
            const int g0 = i0 < I ? i0 : I - 4, g1 = i1 < I ? i1 : I - 4;
            // bias of the lane's 2 x 4 output items: the tile's 32 values through the scalar unit (I % 8 == 0: 32-B aligned), picked by bit masks
            // (?: on scalar-loaded values is turned into branches)
            const int bbase = __builtin_amdgcn_readfirstlane(min(t * 32, I - 32));
            typedef float st2_f8 __attribute__((ext_vector_type(8)));
            const st2_f8* __restrict__ bs = reinterpret_cast<const st2_f8*>(bp1 + bbase);
            const st2_f8 bq0 = bs[0], bq1 = bs[1], bq2 = bs[2], bq3 = bs[3];
            auto pick4 = [&](int g) -> float4 {
                const int sel = (g - bbase) >> 2;      // 0 .. 7: which group of four
                unsigned km[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) km[k] = 0u - (unsigned)(sel == k);
                float4 r;
                float* rp = &r.x;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rp[j] = __uint_as_float((__float_as_uint(bq0[j]) & km[0]) | (__float_as_uint(bq0[4 + j]) & km[1]) | (__float_as_uint(bq1[j]) & km[2]) |
                                            (__float_as_uint(bq1[4 + j]) & km[3]) | (__float_as_uint(bq2[j]) & km[4]) | (__float_as_uint(bq2[4 + j]) & km[5]) |
                                            (__float_as_uint(bq3[j]) & km[6]) | (__float_as_uint(bq3[4 + j]) & km[7]));
                return r;
            };
            const float4 bias0 = pick4(g0), bias1 = pick4(g1);
            ltg_f32x4 acc[2][NTB];
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) {
                acc[0][nt] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
                acc[1][nt] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < ST_KS; ++ks) {
                ltg_u32x4 bfr[NTB];
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) bfr[nt] = Hl[(ks * NTB + nt) * 64];
                const ltg_bf16x8 a0 = __builtin_bit_cast(ltg_bf16x8, Wr[(2 * ks) % ST2_RING]);
                const ltg_bf16x8 a1 = __builtin_bit_cast(ltg_bf16x8, Wr[(2 * ks + 1) % ST2_RING]);
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) {
                    const ltg_bf16x8 b = __builtin_bit_cast(ltg_bf16x8, bfr[nt]);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[1][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // refill: units 2 ks + 19 and 2 ks + 20 (this tile's second half, then the next tile's first half) into the registers just consumed
                if (2 * ks + ST2_RING < ST2_UNITS) { ST2_LOAD(2 * ks + ST2_RING, c0, c1) } else { ST2_LOAD(2 * ks + ST2_RING - ST2_UNITS, n0, n1) }
                if (2 * ks + 1 + ST2_RING < ST2_UNITS) { ST2_LOAD(2 * ks + 1 + ST2_RING, c0, c1) } else { ST2_LOAD(2 * ks + 1 + ST2_RING - ST2_UNITS, n0, n1) }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: + bias, 16-B stores (row 16 nt + lr, items g .. g + 3), running softmax statistics of the row over this lane's items
            const bool in0 = i0 < I, in1 = i1 < I;
            int lr_v = lr;      // (opaque: hoisted out of the loop the NTB row offsets cost registers the ring needs)
            asm volatile("" : "+v"(lr_v));
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) {
                const unsigned ro = (unsigned)min(16 * nt + lr_v, M - 1) * (unsigned)I;
                const ltg_f32x4 v0{acc[0][nt][0] + bias0.x, acc[0][nt][1] + bias0.y, acc[0][nt][2] + bias0.z, acc[0][nt][3] + bias0.w};
                const ltg_f32x4 v1{acc[1][nt][0] + bias1.x, acc[1][nt][1] + bias1.y, acc[1][nt][2] + bias1.z, acc[1][nt][3] + bias1.w};
                *(ltg_f32x4 __attribute__((address_space(1)))*)(Lg + (ro + (unsigned)g0) * 4u) = v0;
                *(ltg_f32x4 __attribute__((address_space(1)))*)(Lg + (ro + (unsigned)g1) * 4u) = v1;
                if constexpr (STATS) {
                    const float m0 = in0 ? fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v0[2], v0[3])) : -INFINITY;
                    const float m1 = in1 ? fmaxf(fmaxf(v1[0], v1[1]), fmaxf(v1[2], v1[3])) : -INFINITY;
                    const float mn = fmaxf(rm[nt], fmaxf(m0, m1)), mr = fmaxf(mn, -1e30f);
                    float e = rs[nt] * __expf(rm[nt] - mr);
                    const float e0 = (__expf(v0[0] - mr) + __expf(v0[1] - mr)) + (__expf(v0[2] - mr) + __expf(v0[3] - mr));
                    const float e1 = (__expf(v1[0] - mr) + __expf(v1[1] - mr)) + (__expf(v1[2] - mr) + __expf(v1[3] - mr));
                    e += in0 ? e0 : 0.f;
                    e += in1 ? e1 : 0.f;
                    rs[nt] = e;
                    rm[nt] = mn;
                }
            }
            c0 = n0;
            c1 = n1;
        }
    }
#undef ST2_LOAD
    if constexpr (STATS) {
        // the four lane groups of a row, then the eight waves through LDS (the h2 image is dead: one barrier in front)
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float m2 = __shfl_xor(rm[nt], o), s2 = __shfl_xor(rs[nt], o);
                const float mn = fmaxf(rm[nt], m2), mr = fmaxf(mn, -1e30f);
                rs[nt] = rs[nt] * __expf(rm[nt] - mr) + s2 * __expf(m2 - mr);
                rm[nt] = mn;
            }
        }
        __syncthreads();
        float2* red = reinterpret_cast<float2*>(Hs);   // [8 waves][NTB * 16 rows]
        if (lq == 0) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) red[w * (NTB * 16) + 16 * nt + lr] = make_float2(rm[nt], rs[nt]);
        }
        __syncthreads();
        if (tid < M) {
            float m = -INFINITY, sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float2 q = red[ww * (NTB * 16) + tid];
                const float mn = fmaxf(m, q.x), mr = fmaxf(mn, -1e30f);
                sum = sum * __expf(m - mr) + q.y * __expf(q.x - mr);
                m = mn;
            }
            float* o2 = stat + ((size_t)blockIdx.x * M + tid) * 2;
            o2[0] = m;
            o2[1] = sum;
        }
    }
}

// part[blockIdx.x][b][h]: this workgroup's share of dh2 (k_da2 sums the slabs)
// NH = 2: blockIdx.y = which HALF of the 608 columns this workgroup produces, over a chunk of twice the items -- the same number of
// workgroups and the same W bytes per workgroup, but half the partial slabs (at 25 024 items 98 x 240 KB instead of 196: the slab sum
// that follows on the critical stream reads 23.5 MB instead of 47) and half the accumulator registers (76 instead of 152)
template <bool D16, int NH = 1>
__global__ __launch_bounds__(ST_NT) void k_dh2_stream(int B, int I, int H, int chunk, const float* __restrict__ dlog,
                                                      const unsigned short* __restrict__ Wb, float* __restrict__ part,
                                                      LtgGate started = LTG_NO_GATE) {
    extern __shared__ __attribute__((aligned(16))) unsigned short st_lds[];  // 2 x [ST_BN][ST_LDW]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    // started: opened by the first workgroup as soon as this kernel runs -- dlogits is complete.  With TWO shadow buffers (ltg_pipe.shadow_out)
    // that is all the forked weight update waits for: it writes the other buffer while this product reads Wb.
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) ltg_gate_set(started);
    constexpr int NTL = ST_KP / 16 / NH;  // 38 (19) column tiles of the accumulator
    const int half = NH == 2 ? (int)blockIdx.y : 0;
    auto fetch_w = [&](int i0_, StW& r_) {
        if constexpr (NH == 2) st_fetch_w_half(Wb, I, i0_, half, r_);
        else st_fetch_w(Wb, I, i0_, r_);
    };
    auto stash_w = [&](unsigned short* Wl_, const StW& r_) {
        if constexpr (NH == 2) st_stash_w_half(Wl_, r_);
        else st_stash_w(Wl_, r_);
    };
    ltg_f32x4 acc[NTL];
#pragma unroll
    for (int n = 0; n < NTL; ++n) acc[n] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int ibeg = blockIdx.x * chunk, iend = min(I, ibeg + chunk);
    if (ibeg >= iend) return;
    const int row = 16 * w + lr;
    const float* drow = dlog + (size_t)min(row, B - 1) * I;
    const unsigned short* drow16 = reinterpret_cast<const unsigned short*>(dlog) + (size_t)min(row, B - 1) * I;   // D16: [B][I] bf16
    const int ilast = ibeg + (iend - ibeg - 1) / ST_BN * ST_BN;      // start of the chunk's last 32-item step
    // A fragment of a 32-item step: dlog[row][i0 + 8*lq .. +7] (I % 8 == 0: whole 32-B groups, 16-B aligned).  Loads are
    // unconditional (clamped); rows >= B and steps past the end of the chunk are zeroed with a bit mask, so a clamped
    // duplicate step adds nothing.  No branch in the loop: every s_waitcnt is an exact count.
#define DH_LOAD_A(i0, X0, X1)                                                      \
    {                                                                              \
        const int ib_ = min(min((i0), ilast) + 8 * lq, I - 8);                     \
        if constexpr (D16) X0 = *reinterpret_cast<const float4*>(drow16 + ib_);    \
        else {                                                                     \
            X0 = *reinterpret_cast<const float4*>(drow + ib_);                     \
            X1 = *reinterpret_cast<const float4*>(drow + ib_ + 4);                 \
        }                                                                          \
    }
    StW r0, r1;
    float4 e0, e1, o0, o1;     // A fragments of the even / odd steps
    // (round 5: all four requests of the prologue first, THEN the first tile's way into LDS -- stashed right behind its own request it made
    // the prologue two dependent round trips, in workgroups whose whole chunk is five or six steps)
    fetch_w(ibeg, r0);
    DH_LOAD_A(ibeg, e0, e1)
    fetch_w(min(ibeg + ST_BN, ilast), r1);
    DH_LOAD_A(ibeg + ST_BN, o0, o1)
    __builtin_amdgcn_sched_barrier(0);
    stash_w(st_lds, r0);
    __syncthreads();
    int cur = 0;
    const int tq = lr >> 2, tp = lr & 3;
    // DH_STEP(RL, RS, X0, X1): LDS[cur] = W tile of step i0, RS = W tile of the next step (in flight), X = A fragment of
    // step i0 (requested two steps ago); requests the W tile two steps ahead into RL and, once X is converted, the A
    // fragment two steps ahead into X again -- two steps of HBM loads are always outstanding.
#define DH_STEP(RL, RS, X0, X1)                                                                                                 \
    {                                                                                                                           \
        fetch_w(min(i0 + 2 * ST_BN, ilast), RL);                                                                                \
        const unsigned keep = (row < B && i0 + 8 * lq < iend) ? 0xFFFFFFFFu : 0u;                                               \
        ltg_u32x4 au;                                                                                                           \
        if constexpr (D16) {                                                                                                    \
            au[0] = __float_as_uint(X0.x) & keep; au[1] = __float_as_uint(X0.y) & keep;                                         \
            au[2] = __float_as_uint(X0.z) & keep; au[3] = __float_as_uint(X0.w) & keep;                                         \
        } else {                                                                                                                \
            const uint2 pa = ltg_pack4(X0), pb = ltg_pack4(X1);                                                                 \
            au[0] = pa.x & keep; au[1] = pa.y & keep; au[2] = pb.x & keep; au[3] = pb.y & keep;                                 \
        }                                                                                                                       \
        const ltg_bf16x8 af = __builtin_bit_cast(ltg_bf16x8, au);                                                               \
        DH_LOAD_A(i0 + 2 * ST_BN, X0, X1)                                                                                       \
        const unsigned short* Wl = st_lds + cur * ST_BN * ST_LDW;                                                               \
        /* transposed fragment reads: lane 4q+p of each 16-lane group addresses row (8*lq + q), columns 4p..4p+3; it */        \
        /* receives column (lane & 15) of the four rows -> k = 8*lq + q (first read), 8*lq + 4 + q (second) */                  \
        const unsigned short* tbase = Wl + (8 * lq + tq) * ST_LDW + 4 * tp;                                                     \
        _Pragma("unroll") for (int n = 0; n < NTL; ++n) {                                                                       \
            typedef ltg_s16x4 __attribute__((address_space(3))) * lds_p;                                                        \
            const ltg_s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tbase + n * 16));                              \
            const ltg_s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tbase + 4 * ST_LDW + n * 16));                 \
            ltg_u16x8 bu;                                                                                                       \
            bu[0] = b0[0]; bu[1] = b0[1]; bu[2] = b0[2]; bu[3] = b0[3];                                                         \
            bu[4] = b1[0]; bu[5] = b1[1]; bu[6] = b1[2]; bu[7] = b1[3];                                                         \
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(ltg_bf16x8, bu), acc[n], 0, 0, 0);          \
        }                                                                                                                       \
        stash_w(st_lds + (cur ^ 1) * ST_BN * ST_LDW, RS);                                                                       \
        __syncthreads();                                                                                                        \
        cur ^= 1;                                                                                                               \
        i0 += ST_BN;                                                                                                            \
    }
    for (int i0 = ibeg; i0 < iend;) {
        DH_STEP(r0, r1, e0, e1)
        DH_STEP(r1, r0, o0, o1)
    }
#undef DH_STEP
#undef DH_LOAD_A
    float* out = part + (size_t)blockIdx.x * B * H;
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
        const int h = half * (ST_KP / 2) + n * 16 + lr;
        if (h < H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = 16 * w + 4 * lq + q;
                if (rr < B) out[(size_t)rr * H + h] = acc[n][q];
            }
        }
    }
}

// dW_p1t[i][h] = sum_b dlog[b][i] * h2[b][h] (+ ones column h == H -> db_p1[i]) fused with the Adam update of
// W_p1t / b_p1 and the refresh of the bf16 shadow, streaming: persistent 8-wave workgroups, h2 fragments
// stationary in registers (wave w owns columns [80w, 80w+80)), dlog read ONCE in [128 rows][32 items] tiles
// (row-major bf16 LDS image, consumed transposed by ds_read_b64_tr_b16), theta/m/v touched exactly once.
constexpr int DW_LDD = 40;  // LDS row stride of the dlog tile in bf16 (80 B: 16-B aligned)
constexpr int DW_LDC = 84;  // row stride of a wave's fp32 gradient block (80 columns + 4)
// read_h2: the one-call step's hand-over of h2 (ltg_pipe.sync words 8 / 1).  h2 is read in the prologue only; a workgroup that has its
// fragments counts itself in word 8, and the one that completes the grid stores the call's ordinal into word 1 -- from then on the next
// step's dec-0 may overwrite h2 although this kernel still runs (a write-after-read hazard: the reads have returned, nothing is published).
struct LtgH2Done {
    unsigned* count;   // NULL: no hand-over
    unsigned* word;
    unsigned seq;
    const unsigned* poison;
};
template <bool D16>
__global__ __launch_bounds__(ST_NT) void k_dec1_bwd_adam_stream(int B, int I, int H, const float* __restrict__ dlog,
                                                                const float* __restrict__ h2, ltg_gen_state st, AdamC ad,
                                                                LtgH2Done hd = LtgH2Done{nullptr, nullptr, 0u, nullptr}) {
    __shared__ __attribute__((aligned(16))) unsigned short Dl[2][128 * DW_LDD];
    __shared__ __attribute__((aligned(16))) float Cs[8 * 32 * DW_LDC];   // per-wave [32][80] gradient blocks
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    if (ltg_poisoned(hd.poison)) return;
    float4 *W4 = reinterpret_cast<float4*>(st.p[3]), *M4 = reinterpret_cast<float4*>(st.m[3]), *V4 = reinterpret_cast<float4*>(st.v[3]);
#if defined(LTG_X_SPIN)
    // MEASUREMENT BUILD ONLY (results wrong): the kernel's footprint (registers, LDS, one workgroup per CU) for LTG_X_SPIN us, no memory traffic
    {
        Dl[0][tid] = 0;
        Cs[tid] = 0.f;
        asm volatile("v_mov_b32 v220, 0" ::: "v220");
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)(LTG_X_SPIN) * 100ull) __builtin_amdgcn_s_sleep(16);
        if (Dl[0][tid] == 1) W4[0].x = Cs[tid];
        return;
    }
#elif defined(LTG_X_NOUPDATE)
    // MEASUREMENT BUILD ONLY (weights do not move): no update at all -- what the chain costs with nothing beside it
    if (B >= 0) return;
#endif
    float *bb = st.p[7], *mb = st.m[7], *vb = st.v[7];
    unsigned short* Wb = st.wp1t_bf16;
    // stationary B fragments: B[k = b][n] = h2[b][n] (n < H), 1 (n == H), 0 beyond.  Built through LDS in four
    // 32-row slices (coalesced float4 reads, bf16 image with a conflict-free 650-element row stride); gathering the
    // 160 values of a lane straight from global memory makes the compiler hoist 160 loads and spill the fragments.
    ltg_bf16x8 bf[5][4];
    {
        constexpr int HS = 650;
        unsigned short* Hs = reinterpret_cast<unsigned short*>(Cs);   // [32][HS] bf16 = 41.6 KB of the 86 KB slab area
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            for (int e = tid; e < 32 * 160; e += ST_NT) {
                const int rr = e / 160, n = 4 * (e % 160), b = ks * 32 + rr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < B && n < H) v = *reinterpret_cast<const float4*>(h2 + (size_t)b * H + n);
                uint2 pk = ltg_pack4(v);
                if (b < B && n == H) pk.x = 0x3F80u;                  // the ones column
                unsigned* dst = reinterpret_cast<unsigned*>(Hs + rr * HS + n);
                dst[0] = pk.x;
                dst[1] = pk.y;
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const int n = 80 * w + 16 * nt + lr;
                ltg_u16x8 t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = Hs[(8 * lq + j) * HS + n];
                bf[nt][ks] = __builtin_bit_cast(ltg_bf16x8, t);
            }
            __syncthreads();
        }
    }
    if (hd.count && tid == 0) {   // (behind the prologue's last barrier: every h2 value this workgroup needs sits in registers)
        if (atomicAdd(hd.count, 1u) + 1u == gridDim.x) {
            __hip_atomic_store(hd.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next call's kernel starts behind this one on its stream)
            __hip_atomic_store(hd.word, hd.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int ntiles = I / 32, G = gridDim.x;   // full tiles only: the host sends the ragged tail (I % 32 rows) to k_dec1_bwd_adam
    // dlog tile loader: thread -> (row b = tid / 4, 8 items at 8 * (tid % 4))
    const int lb = tid >> 2, lseg = tid & 3;
    const float* lrow = dlog + (size_t)min(lb, B - 1) * I;
    const unsigned short* lrow16 = reinterpret_cast<const unsigned short*>(dlog) + (size_t)min(lb, B - 1) * I;   // D16: [B][I] bf16
    auto fetch = [&](int t, float4& x0, float4& x1) {
        const int ib = min(t * 32 + 8 * lseg, I - 8);
        if constexpr (D16) x0 = *reinterpret_cast<const float4*>(lrow16 + ib);   // eight bf16 = the LDS image as it is
        else {
            x0 = *reinterpret_cast<const float4*>(lrow + ib);
            x1 = *reinterpret_cast<const float4*>(lrow + ib + 4);
        }
    };
    auto stash = [&](unsigned short* D, int t, const float4& x0, const float4& x1) {
        const bool ok = lb < B && t * 32 + 8 * lseg < I;
        ltg_u32x4 p;
        if constexpr (D16) {
            p[0] = ok ? __float_as_uint(x0.x) : 0u; p[1] = ok ? __float_as_uint(x0.y) : 0u;
            p[2] = ok ? __float_as_uint(x0.z) : 0u; p[3] = ok ? __float_as_uint(x0.w) : 0u;
        } else {
            const uint2 a = ltg_pack4(x0), c = ltg_pack4(x1);
            p[0] = ok ? a.x : 0u; p[1] = ok ? a.y : 0u; p[2] = ok ? c.x : 0u; p[3] = ok ? c.y : 0u;
        }
        *reinterpret_cast<ltg_u32x4*>(D + lb * DW_LDD + 8 * lseg) = p;
    };
    int t = blockIdx.x, cur = 0;
    float4 x0, x1;
    if (t < ntiles) {
        fetch(t, x0, x1);
        stash(Dl[0], t, x0, x1);
        fetch(t + G < ntiles ? t + G : t, x0, x1);
    }
    __syncthreads();
    const int tq = lr >> 2, tp = lr & 3;
    // ---- Adam epilogue geometry.  The 32 rows of a tile are CONTIGUOUS in theta / m / v (32 x H floats, the tile starts on
    // a 256-B boundary), so ownership is by rows, not by the column blocks the products were computed in: wave w walks
    // rows 4w..4w+3 = 4 H/4 consecutive float4 as ten 1-KiB wave accesses (float4 e = 64 jj + lane; the tail lanes of
    // the tenth access mirror the last element: same load, same result, same store).  The gradient of element (row,
    // chunk) is fetched from the LDS slab of whichever wave computed that column block.  No lane predicates, no
    // wave-dependent branches: every s_waitcnt in the loop is an exact count.  theta / m / v travel in six
    // software-pipelined stages (2+2+2+2+1+1 float4 per lane) rotating over three register sets; the loads of the stage
    // after next -- at the end of a tile: of the NEXT tile's first two stages -- are issued before the current stage is
    // consumed, so HBM requests stay in flight through the MFMA phase and the barriers.  (A fourth set does not fit: the
    // stationary fragments hold 80 of the 256 VGPRs, the kernel uses 248.)
    float* Cw = Cs + w * (32 * DW_LDC);                        // this wave's product slab
    const int H4 = H >> 2, nel = 4 * H4;                       // float4 per row / per wave and tile (host: 9 * 64 < nel <= 10 * 64)
    const unsigned rowB = (unsigned)H * 4u;
    const unsigned lo = 16u * lane, lo9 = 16u * (unsigned)(min(576 + lane, nel - 1) - 576);
    ltg_f32x4 Ap[2], Am[2], Av[2], Bp[2], Bm[2], Bv[2], Cp[2], Cm[2], Cv[2];
#define DW_ADAM1(f)                                  \
    {                                                \
        float p_ = p.f, m_ = mm.f, v_ = v2.f;        \
        adam1(p_, m_, v_, g.f, ad.lr_t, ad);         \
        p.f = p_; mm.f = m_; v2.f = v_;              \
    }
    // addressing: (uniform row base, computed on the scalar unit) + (one 32-bit lane offset per pass) -- global_load with
    // an SGPR base, so the unrolled stages do not pin a VGPR pair per access
#define DW_AT(T, BASE, UB, LB) (*(T __attribute__((address_space(1)))*)(ltg_uniform_ptr(reinterpret_cast<const char*>(BASE) + (UB)) + (LB)))
    // theta / m / v as NON-TEMPORAL accesses (the nt bit of global_load / global_store): each element is touched exactly once per step, by
    // this kernel only (the forward reads the bf16 shadow) -- 360 MB per step at 25 024 items that would otherwise push everything else
    // out of the L2s and the memory-side cache.  Measured (same box, interleaved): per-rank proxy 197-204 -> 194 us per G step, C4-shaped
    // phase G 272-274 -> 256-262 ms, C3-shaped 117.6 -> 114.7 ms.  -DLTG_DW_TEMPORAL builds the plain accesses.
#ifndef LTG_DW_TEMPORAL
#define DW_LDG(P) __builtin_nontemporal_load(P)
#define DW_STG(V, P) __builtin_nontemporal_store(V, P)
#else
#define DW_LDG(P) (*(P))
#define DW_STG(V, P) (*(P) = (V))
#endif
#define DW_LD(S, tt, NJ, J0, LOFF)                                                      \
    _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                 \
        const size_t u = (size_t)((tt) * 32 + 4 * w) * rowB + 1024u * ((J0) + jj);      \
        S##p[jj] = DW_LDG(&DW_AT(const ltg_f32x4, W4, u, LOFF));                        \
        S##m[jj] = DW_LDG(&DW_AT(const ltg_f32x4, M4, u, LOFF));                        \
        S##v[jj] = DW_LDG(&DW_AT(const ltg_f32x4, V4, u, LOFF));                        \
    }
#define DW_AP(S, tt, NJ, J0, LOFF)                                                      \
    _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                 \
        const int e = min(64 * ((J0) + jj) + lane_v, nel - 1);                          \
        const int rl = e / H4, ch = e - rl * H4, wb = ch / 20, lc = ch - 20 * wb;       \
        const float4 g = *reinterpret_cast<const float4*>(Cs + wb * (32 * DW_LDC) + (4 * w + rl) * DW_LDC + 4 * lc); \
        ltg_f32x4 p = S##p[jj], mm = S##m[jj], v2 = S##v[jj];                           \
        DW_ADAM1(x) DW_ADAM1(y) DW_ADAM1(z) DW_ADAM1(w)                                 \
        const size_t u = (size_t)((tt) * 32 + 4 * w) * rowB + 1024u * ((J0) + jj);      \
        DW_STG(p, &DW_AT(ltg_f32x4, W4, u, LOFF));                                      \
        DW_STG(mm, &DW_AT(ltg_f32x4, M4, u, LOFF));                                     \
        DW_STG(v2, &DW_AT(ltg_f32x4, V4, u, LOFF));                                     \
        const uint2 pk = ltg_pack4(make_float4(p.x, p.y, p.z, p.w));                    \
        DW_AT(ltg_u32x2, Wb, (size_t)((tt) * 32 + 4 * w) * (ST_KP * 2), (unsigned)(rl * (ST_KP * 2) + 8 * ch)) = ltg_u32x2{pk.x, pk.y}; \
    }
    // THREE register sets rotate over the six stages (A B C A B C: the next tile starts on A again), so two stages of loads are in flight
    // behind the one being consumed and the next tile's first two stages through its MFMA phase: 248 VGPRs, no scratch.  Against two
    // sets (round 4, interleaved, two boxes): 200 000 items 741-774 against 755-806 us per step, 20 000 and 25 024 items equal
    // (138.0-142.6 / 137.5-139.2, 151.8-153.7 / 151.4-155.0).  (Two sets that only keep BOTH loaded through the MFMA phase: equal everywhere.)
#define DW_SB __builtin_amdgcn_sched_barrier(0);
#define DW_STAGES() \
        DW_LD(C, t, 2, 4, lo) DW_SB   DW_AP(A, t, 2, 0, lo) DW_SB \
        DW_LD(A, t, 2, 6, lo) DW_SB   DW_AP(B, t, 2, 2, lo) DW_SB \
        DW_LD(B, t, 1, 8, lo) DW_SB   DW_AP(C, t, 2, 4, lo) DW_SB \
        DW_LD(C, t, 1, 9, lo9) DW_SB  DW_AP(A, t, 2, 6, lo) DW_SB \
        DW_LD(A, tn, 2, 0, lo) DW_SB  DW_AP(B, t, 1, 8, lo) DW_SB   /* the next tile's first two stages (of this one again at the end: unused) */ \
        DW_LD(B, tn, 2, 2, lo) DW_SB  DW_AP(C, t, 1, 9, lo9) DW_SB
#define DW_FIRST(tt) DW_LD(A, tt, 2, 0, lo) DW_LD(B, tt, 2, 2, lo)
    // one tile: sets A and B hold its first two stages (requested at the end of the previous tile) -- one loop body, no register-set swap
    // (a swap would have to wait for loads in flight)
#define DW_BODY()                                                                                              \
    {                                                                                                                \
        const bool more = t + G < ntiles;                                                                            \
        const int tn = more ? t + G : t;                                                                             \
        int lane_v = lane; /* opaque per tile: keeps the ten (row, chunk) -> LDS / shadow offsets of a lane from being   \
                              hoisted out of the loop into 20+ VGPRs (they are a handful of VALU ops to recompute) */  \
        asm volatile("" : "+v"(lane_v));                                                                           \
        ltg_f32x4 acc[2][5];                                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                             \
            _Pragma("unroll") for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = ltg_f32x4{0.f, 0.f, 0.f, 0.f};            \
        const unsigned short* D = Dl[cur];                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                           \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                       \
                typedef ltg_s16x4 __attribute__((address_space(3))) * lds_p;                                         \
                const unsigned short* base = D + (ks * 32 + 8 * lq + tq) * DW_LDD + mt * 16 + 4 * tp;                \
                const ltg_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)base);                           \
                const ltg_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * DW_LDD));            \
                ltg_u16x8 au;                                                                                        \
                au[0] = a0[0]; au[1] = a0[1]; au[2] = a0[2]; au[3] = a0[3];                                          \
                au[4] = a1[0]; au[5] = a1[1]; au[6] = a1[2]; au[7] = a1[3];                                          \
                const ltg_bf16x8 af = __builtin_bit_cast(ltg_bf16x8, au);                                            \
                _Pragma("unroll") for (int nt = 0; nt < 5; ++nt)                                                     \
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nt][ks], acc[mt][nt], 0, 0, 0);     \
            }                                                                                                        \
        }                                                                                                            \
        /* the wave's gradient block takes a round trip through its private LDS slab (MFMA C layout -> row chunks) */ \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                             \
            _Pragma("unroll") for (int nt = 0; nt < 5; ++nt)                                                         \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) Cw[(mt * 16 + 4 * lq + q) * DW_LDC + 16 * nt + lr] = acc[mt][nt][q]; \
        __syncthreads();                                                                                             \
        /* dlog: x (tile t + G, requested one tile ago) -> the idle LDS buffer; request tile t + 2G.  Both             \
           unconditional (clamped): a branch around either ends in vmcnt(0) at its join. */                          \
        stash(Dl[cur ^ 1], more ? t + G : t, x0, x1);                                                                \
        fetch(t + 2 * G < ntiles ? t + 2 * G : t, x0, x1);                                                           \
        /* bias b_p1 of rows 4w .. 4w+3 (gradient = the ones column, column H of the product): lanes >= 4 mirror    \
           lane 3, loads here, update at the end of the tile -- no predicate, nothing waits for these loads */       \
        const int ib = t * 32 + 4 * w + min(lane, 3);                                                                \
        float pbv = bb[ib], mbv = mb[ib], vbv = vb[ib];                                                              \
        DW_STAGES()                                                                                                  \
        {                                                                                                            \
            const float gbias = Cs[(H / 80) * (32 * DW_LDC) + (4 * w + min(lane, 3)) * DW_LDC + H % 80];             \
            adam1(pbv, mbv, vbv, gbias, ad.lr_t, ad);                                                                \
            mb[ib] = mbv;                                                                                            \
            vb[ib] = vbv;                                                                                            \
            bb[ib] = pbv;                                                                                            \
        }                                                                                                            \
        __syncthreads();                                                                                             \
        cur ^= 1;                                                                                                    \
    }
    if (t < ntiles) { DW_FIRST(t) }
    for (; t < ntiles; t += G) DW_BODY()
#undef DW_STAGES
#undef DW_SB
#undef DW_FIRST
#undef DW_LDG
#undef DW_STG
#undef DW_AT
#undef DW_LD
#undef DW_AP
#undef DW_ADAM1
#undef DW_BODY
}

// out[i] = the e4m3 value the fp8 GEMM mode stores for in[i] (verification helper: pins the oracle's rounding model)
__global__ __launch_bounds__(NT) void k_fp8_roundtrip(int n, const float* __restrict__ in, float* __restrict__ out) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT)
        out[i] = __builtin_amdgcn_cvt_f32_fp8((int)ltg_f2fp8(in[i]), 0);
}

// C[M][N] = A[M][K] . B[K][N] through the block template in one of its operand modes (verification helper)
template <int MODE>
__global__ __launch_bounds__(NT) void k_debug_gemm(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ Bm,
                                                   float* __restrict__ Cm) {
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    auto a = [=] __device__(int m, int k) -> float { return A[(size_t)m * K + k]; };
    auto b = [=] __device__(int k, int n) -> float { return Bm[(size_t)k * N + n]; };
    auto epi = [=] __device__(int m, int n, float acc) { Cm[(size_t)m * N + n] = acc; };
    ltg_gemm_block<MODE, 32, 32, 128, 2, 2, false, true, false, 4, 4>(M, N, m0, n0, 0, K, a, b, epi);
}

// (re)build the bf16 shadow of W_p1t from the fp32 master rows (set-up / after loading weights)
__global__ __launch_bounds__(NT) void k_refresh_shadow(int I, int H, const float* __restrict__ W, unsigned short* __restrict__ Wb) {
    const size_t total = (size_t)I * ST_KP;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const size_t i = e / ST_KP;
        const int k = (int)(e % ST_KP);
        Wb[e] = k < H ? ltg_f2bf(W[i * H + k]) : (unsigned short)0;
    }
}

// row log-sum-exp of the logits (log_softmax / softmax, MultiVAE.py:108,143)
__global__ __launch_bounds__(NT) void k_row_lse(int I, const float* __restrict__ logits, float* __restrict__ lse) {
    __shared__ float red[NT / 64];
    const float* row = logits + (size_t)blockIdx.x * I;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < I; i += NT) mx = fmaxf(mx, row[i]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < I; i += NT) s += expf(row[i] - mx);
    s = block_sum(s, red);
    if (threadIdx.x == 0) lse[blockIdx.x] = mx + logf(s);
}

__global__ __launch_bounds__(NT) void k_softmax_write(int I, const float* __restrict__ logits, const float* __restrict__ lse,
                                                      float* __restrict__ probs) {
    const size_t base = (size_t)blockIdx.y * I;
    const float l = lse[blockIdx.y];
    for (int i = blockIdx.x * NT + threadIdx.x; i < I; i += gridDim.x * NT) probs[base + i] = expf(logits[base + i] - l);
}

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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// the streaming decoder kernels: bf16, large item slab, a training batch (<= 128 rows), H <= 608, 16-B aligned rows
inline bool stream_ok(const ltg_config* cfg, const ltg_gen_state* gen, int rows) {
    return gen->wp1t_bf16 && cfg->precision == LTG_PREC_BF16 && cfg->n_items >= 8192 && (cfg->n_items % 8) == 0 && rows <= 128 && cfg->h_enc <= ST_KP &&
           (cfg->h_enc % 4) == 0 && (cfg->tuning & 15) != 9;
}
// k_dec1_bwd_adam_stream walks the 4 H/4 float4 a wave owns per tile as exactly ten 64-lane accesses
inline bool dw_stream_ok(int H) { return (H % 4) == 0 && H > 576 && H <= 640; }
constexpr int DH2_NH = 2;   // column halves of the streaming dh2 product (k_dh2_stream<.., NH>)
inline int dh2_stream_chunk1(int I);
inline int dh2_stream_chunk(int I) { return DH2_NH * dh2_stream_chunk1(I); }
inline int dh2_stream_chunk1(int I) {
    // items per workgroup of k_dh2_stream = one partial [B][H] slab each: up to 256 workgroups, but at least 4 tiles per
    // workgroup -- at 20 000 items 157 slabs instead of 209 (each slab is 240 KB written and read once more by k_da2)
    int c = (I + 255) / 256;
    c = (c + ST_BN - 1) / ST_BN * ST_BN;
    return c < 4 * ST_BN ? 4 * ST_BN : c;
}

inline int dh2_kchunk(int I) {
    // split the item dimension so that ~128 workgroups x (H/64) share the reduction
    int chunk = (I + 63) / 64;
    chunk = (chunk + 31) / 32 * 32;
    if (chunk < 256) chunk = 256;
    return chunk;
}

// rows of the sparse W_q0 gradient = distinct items of a batch <= min(n_items, nnz of the batch).  The workspace is sized
// without knowing nnz, so the table bound is used: one heavy user in a short batch can never overflow it.
inline size_t gq0_rows(const ltg_config* cfg, int /*max_rows*/) { return (size_t)cfg->n_items; }

constexpr int RD_MAXI = 4096;   // "small item slab": a row of logits fits the registers of one workgroup (ltg_fast.h)
struct Workspace {
    // generator backward
    float *rowpart, *segpart, *nb, *Pb, *scal, *dlog, *part, *dh2, *da2, *dmlv, *da1, *gq0;
    int32_t* slotmap;   // [I] item -> gradient row of the current batch (only used when the caller passes no slot[] cache)
    // discriminator
    float *A1, *A3, *y, *ds, *lrow, *dpre1, *dpre3, *slab;
    // fast path: per-column-tile partial dot products of the output unit, w4 * dA3/dpre, per-row loss terms of the G step
    float *spart, *G3, *rowout, *xd;
    uint8_t* A1_8;      // the branch layers' output in e4m3 (fp8 operand storage of the wide discriminator)
    // operand-format storage of the fp8 backward (ltg_fp8bwd.h): transposed / row-major e4m3 copies, pair rows padded to np8
    uint8_t *A1T_8, *dpre3_8, *dpre3T_8, *dpre1T_8, *ET_8;
    unsigned short* dA1T_16;
    int np8;
    float* wsp;         // the discriminator's weights split into bf16 terms in MFMA fragment order (ltg_tower.h), rebuilt by every forward-only tower
    size_t bytes;
};
// stride of one discriminator gradient slab: the P gradients + one slot for the chunk's loss sum, padded to whole float4
inline int d_slab_stride(int P) { return (P + 1 + 3) & ~3; }

// geometry of the one-kernel forward-only tower (ltg_tower.h)
constexpr int FT_BM = 64, FT_NT = 512;
constexpr int FT_KP1_MAX = 4;       // h0 <= 128: the gathered embedding rows of a wave stay in registers, split
inline int ft_lda(int h12) { return ((h12 + 31) & ~31) + 4; }      // floats; = 4 mod 32: the 16 rows of a ds_read_b128 fragment read hit every bank once
inline size_t ft_lds_bytes(int h12) { return ((size_t)FT_BM * ft_lda(h12) + 4 * FT_BM) * sizeof(float); }
// fragment triples (3 KiB each) of the three split matrices
inline size_t ft_wsp_triples(int h0, int h1, int h2, int h3) {
    const size_t kp1 = (h0 + 31) / 32, kp3 = (h1 + h2 + 31) / 32;
    return (size_t)((h1 + 15) / 16) * kp1 + (size_t)((h2 + 15) / 16) * kp1 + (size_t)((h3 + 15) / 16) * kp3;
}
inline size_t ft_wsp_bytes(int h0, int h1, int h2, int h3) { return ft_wsp_triples(h0, h1, h2, h3) * 3 * 1024; }
// sizes the one-kernel forward-only tower (ltg_tower.h) serves: the latency path's fp32 discriminator with h0 <= 128 (a wave's gathered rows stay in
// registers), h3 <= 320 (five 16-column tiles per wave column) and A1 [64][h1 + h2] within the LDS.  The choice depends on the layer sizes and
// d_arith only, so the tower inside a step and the batched tower run the same kernel and produce the same bits.
inline bool ft_wsp_capable(const ltg_config* c) {
    return (c->tuning & 262144) == 0 && c->d_precision == LTG_PREC_FP32 && !(c->d_h0 >= 512 && c->d_h1 + c->d_h2 >= 512 && c->d_h3 >= 128) && c->d_h0 >= 4 &&
           c->d_h0 <= 32 * FT_KP1_MAX && (c->d_h0 % 4) == 0 && c->d_h1 >= 1 && c->d_h2 >= 1 && c->d_h3 >= 1 && c->d_h3 <= 320 &&
           ft_lds_bytes(c->d_h1 + c->d_h2) <= (size_t)160 * 1024;
}
Workspace carve(const ltg_config* cfg, int max_rows, int max_pairs, char* base) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = reinterpret_cast<float*>(base + off);
        off += align_up(nfloat * sizeof(float));
        return p;
    };
    const size_t R = (size_t)max_rows, I = (size_t)cfg->n_items, H = (size_t)cfg->h_enc, Z = (size_t)cfg->z_dim;
    const size_t P = (size_t)max_pairs, h12 = (size_t)cfg->d_h1 + cfg->d_h2, h3 = (size_t)cfg->d_h3;
    const int kchunk = dh2_kchunk(cfg->n_items);
    size_t nsplit = (I + kchunk - 1) / kchunk;
    {
        const size_t ns2 = (I + dh2_stream_chunk(cfg->n_items) - 1) / dh2_stream_chunk(cfg->n_items);
        if (ns2 > nsplit) nsplit = ns2;
    }
    w.rowpart = take(R * RP);
    w.segpart = take(segpart_floats(I, R));
    w.nb = take(R);
    w.Pb = take(R);
    w.scal = take(16);
    w.dlog = take(R * I);
    w.part = take(nsplit * R * H);
    w.dh2 = take(R * H);
    w.da2 = take(R * H);
    w.dmlv = take(R * 2 * Z);
    w.da1 = take(R * H);
    w.gq0 = take((size_t)(gq0_rows(cfg, max_rows) + ENC0_BIAS_PARTS) * H);   // sparse gradient rows of W_q0 (+ partial bias rows)
    w.slotmap = reinterpret_cast<int32_t*>(take(I));
    w.A1 = take(P * h12);
    w.A3 = take(P * h3);
    w.y = take(P);
    w.ds = take(P);
    w.lrow = take(P);
    w.dpre1 = take(P * h12);
    w.dpre3 = take(P * h3);
    {
        const DLayout L = d_layout(cfg->d_h0, cfg->d_h1, cfg->d_h2, cfg->d_h3);
        const size_t ks = (P + D_KCHUNK - 1) / D_KCHUNK;
        w.slab = take(ks * (size_t)d_slab_stride(L.off[8]));
    }
    w.spart = take((((h3 + 31) / 32) + 1) * P);   // per-tile partial dot products of the output unit: 32-column tiles, or 2 strips per 64-column tile
    w.G3 = take(P * h3);
    w.rowout = take(R * 4);
    w.xd = take(I <= (size_t)RD_MAXI ? R * I : 1);   // dense operand rows of enc-0 (small item slabs)
    w.A1_8 = reinterpret_cast<uint8_t*>(take((P * h12 + 3) / 4));
    {
        const bool f8 = cfg->d_precision == LTG_PREC_FP8;      // (only that mode carries these buffers)
        const size_t np = ((P + 127) / 128 * 128), h0 = (size_t)cfg->d_h0;
        auto take8 = [&](size_t nbytes) { return reinterpret_cast<uint8_t*>(take(f8 ? (nbytes + 3) / 4 : 1)); };
        w.np8 = (int)np;
        w.A1T_8 = take8((h12 + 1) * np);
        w.dA1T_16 = reinterpret_cast<unsigned short*>(take8(2 * h12 * np));   // dA1 / dpre1 of the branch layers, bf16, transposed (ltg_fp8bwd.h)
        w.dpre3_8 = take8(P * h3);
        w.dpre3T_8 = take8(h3 * np);
        w.dpre1T_8 = take8(h12 * np);
        w.ET_8 = take8(2 * (h0 + 1) * np);
    }
    w.wsp = take(ft_wsp_capable(cfg) ? ft_wsp_bytes(cfg->d_h0, cfg->d_h1, cfg->d_h2, cfg->d_h3) / sizeof(float) : 1);
    w.bytes = off;
    return w;
}

inline AdamC make_adam(const ltg_config* cfg, int t) {
    AdamC a;
    const double b1 = cfg->beta1, b2 = cfg->beta2;
    a.lr_t = (float)((double)cfg->lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
    a.b1 = cfg->beta1;
    a.b2 = cfg->beta2;
    a.eps = cfg->adam_eps;
    return a;
}

struct Probe {
    const ltg_probe* p;
    hipStream_t st;
    inline void before(int id) const {
        if (p && p->kernel_id == id && p->ev_start) (void)hipEventRecord((hipEvent_t)p->ev_start, st);
    }
    inline void after(int id) const {
        if (p && p->kernel_id == id && p->ev_stop) (void)hipEventRecord((hipEvent_t)p->ev_stop, st);
    }
};
#define LTG_PROBED(pr, id, stmt) \
    do {                         \
        (pr).before(id);         \
        stmt;                    \
        (pr).after(id);          \
    } while (0)

inline int check_launch() { return hipGetLastError() == hipSuccess ? LTG_OK : LTG_ELAUNCH; }
// hipGetLastError is sticky across unrelated runtime calls of the host program: clear it on entry
inline void clear_errors() { (void)hipGetLastError(); }

inline dim3 grid2(int N, int M, int bn = 64, int bm = 64, int z = 1) { return dim3((N + bn - 1) / bn, (M + bm - 1) / bm, z); }

}  // namespace
namespace {
#include "ltg_fast.h"
#include "ltg_fp8bwd.h"
#include "ltg_tower.h"
}
namespace {

// Which of the round-2 latency-path kernels (ltg_fast.h) apply.  Tuning-knob bit 18 of ltg_config.tuning switches all of
// them off (the round-1 kernels compute the same function; kept for A/B measurements and as the path of unusual sizes).
inline bool fast_on(const ltg_config* c) { return (c->tuning & 262144) == 0; }
inline bool mid_fast(const ltg_config* c, int rows) { return fast_on(c) && (c->z_dim % 4) == 0 && rows <= 256; }
inline bool d_wide(const ltg_config* c) { return c->d_h0 >= 512 && c->d_h1 + c->d_h2 >= 512 && c->d_h3 >= 128; }
// fp8 discriminator with EVERY GEMM operand in operand format (ltg_fp8bwd.h).  Tuning-knob bit 24 (register-resident forward
// tiles) and bit 19 (backward converts on the fly, the round-2 path) switch it off.
inline bool d_fp8_opfmt(const ltg_config* c, const ltg_disc_state* d) {
    return fast_on(c) && c->d_precision == LTG_PREC_FP8 && d->emb_fp8 && d->w1t_fp8 && d->w2t_fp8 && d->w3t_fp8 && d->w3_fp8 && (c->d_h0 % 128) == 0 &&
           ((c->d_h1 + c->d_h2) % 128) == 0 && (c->d_h3 % 128) == 0 && (c->d_h1 % 64) == 0 && (c->d_h2 % 64) == 0 && c->d_h3 <= 64 * D8_OUT_CM &&
           (c->tuning & ((1 << 24) | (1 << 19))) == 0;
}
inline bool d_fast(const ltg_config* c) {
    return fast_on(c) && c->d_precision == LTG_PREC_FP32 && !d_wide(c) && c->d_h3 <= 512 && (c->d_h0 % 4) == 0 && ((c->d_h1 + c->d_h2) % 4) == 0 && (c->d_h3 % 4) == 0;
}
// ltg_config.d_arith: SPL of kernel `which` (0 fk_d_l1, 1 fk_d_l2, 2 fk_d_bwd1, 3 fk_d_bwd2) of the config.ini-sized fp32 discriminator step --
// 0 = v_mfma_f32_16x16x4_f32, 6 / 4 = bf16 cross terms of the split operands (ltg_rgemm.h).  Bits 4-7 choose the kernels (measurements);
// 0 = the library's set LTG_D_SPLIT_SET.
#ifndef LTG_D_SPLIT_SET
#define LTG_D_SPLIT_SET 0xE      // l2, bwd1, bwd2 (l1: one 16 x 16 output per wave -- the split's 36 vector instructions per block buy 4 MFMAs of 32 cycles)
#endif
inline int d_spl(const ltg_config* c, int which) {
    const int mode = c->d_arith & 3, set = ((c->d_arith >> 4) & 15) ? ((c->d_arith >> 4) & 15) : LTG_D_SPLIT_SET;
    if (mode == LTG_DARITH_FP32 || !((set >> which) & 1)) return 0;
    return mode == LTG_DARITH_BF16X4 ? 4 : 6;
}
#define LTG_D_SPL_LAUNCH(SPLV, KERNEL, ...)                                       \
    do {                                                                          \
        if ((SPLV) == 6) hipLaunchKernelGGL((KERNEL<6>), __VA_ARGS__);            \
        else if ((SPLV) == 4) hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<0>), __VA_ARGS__);                        \
    } while (0)
inline bool unsharded(const ltg_config* c) { return c->item_lo == 0 && (c->n_items_global == 0 || c->n_items_global == c->n_items); }
inline bool small_fast(const ltg_config* c, int rows) {
    return fast_on(c) && unsharded(c) && c->n_items <= RD_MAXI && (c->n_items % 4) == 0 && (c->z_dim % 4) == 0 && rows <= 256;
}

bool cfg_ok(const ltg_config* c) {
    return c && c->n_items > 0 && c->h_enc > 0 && c->h_enc <= 768 && (c->h_enc % 4) == 0 && c->z_dim > 0 &&
           (c->precision == LTG_PREC_BF16 || c->precision == LTG_PREC_FP32) && c->d_precision >= 0 && c->d_precision <= LTG_PREC_FP8 &&
           c->d_arith >= 0 && (c->d_arith & 3) <= LTG_DARITH_BF16X4 && (c->d_arith & ~0xF3) == 0;
}

inline int Ig_of(const ltg_config* cfg) { return cfg->n_items_global > 0 ? cfg->n_items_global : cfg->n_items; }

// lazy Adam clock of W_q0 (ltg_gen_state.q0_last): usable when the caller supplies clock, history ring and a period
inline bool q0_lazy(const ltg_config* cfg, const ltg_gen_state* gen) {
    return gen->q0_last && gen->q0_lr_hist && gen->q0_period >= 1 && gen->q0_period <= LTG_Q0_HIST / 2 && gen->q0_ord >= 0 && (cfg->h_enc % 4) == 0 &&
           cfg->n_items >= 8192;   // smaller slabs update W_q0 as a dense product: nothing to defer
}
// the item rows this batch reads, up to the caller's clock (no-ops for rows that are current)
void q0_touch(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, hipStream_t st, const unsigned* poison = nullptr,
              int32_t* mark = nullptr, unsigned seq = 0u) {
    if (!q0_lazy(cfg, gen) || bt->n_rows <= 0) return;
    const AdamC ad = make_adam(cfg, 1);   // b1, b2, eps; the learning rates come from the history ring
    if (bt->uptr && bt->csr_pos) {
        if (bt->n_unique > 0)
            hipLaunchKernelGGL(k_q0_touch_unique, dim3(bt->n_unique), dim3(Q0_NT), 0, st, cfg->h_enc, bt->n_unique, bt->uptr, bt->csr_pos, bt->indices, bt->uitem,
                               gen->q0_ord, *gen, ad, poison, mark, seq);
    } else {
        hipLaunchKernelGGL(k_q0_touch_rows, dim3(bt->n_rows), dim3(Q0_NT), 0, st, cfg->h_enc, bt->n_rows, bt->indptr, bt->indices, gen->q0_ord, *gen, ad);
    }
}

// stage 1: enc-0 over this rank's item slab.  pre_only: leave the partial pre-activation in acts->h1.
void fwd_stage_enc(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_fwd_opts* o,
                   const ltg_gen_acts* acts, int pre_only, hipStream_t st, float* xd = nullptr, bool touched = false,
                   LtgGate started = LTG_NO_GATE, LtgGate end_wait = LTG_NO_GATE) {
    const int R = bt->n_rows, I = cfg->n_items, H = cfg->h_enc;
    const Probe pr{o->probe, st};
    if (!touched) q0_touch(cfg, gen, bt, st);
    if (fast_on(cfg)) {
        LTG_PROBED(pr, LTG_K_ENC0_FWD,
                   hipLaunchKernelGGL(fk_enc0_fwd, dim3((H / 4 + 63) / 64, R + (end_wait.word ? 1 : 0)), dim3(ENC_NT), xd ? (size_t)I * sizeof(float) : 0, st, H, I, bt->indptr,
                                      bt->indices, bt->values, o->drop_keep, o->keep_prob, cfg->seed, o->rng_step, gen->p[0], gen->p[4], acts->h1,
                                      acts->row_scale, bt->row_norm2, cfg->item_lo, Ig_of(cfg), pre_only, xd, o->rows_per_step, started, end_wait));
        return;
    }
    LTG_PROBED(pr, LTG_K_ENC0_FWD,
               hipLaunchKernelGGL(k_enc0_fwd, dim3(R), dim3(ENC_NT), (size_t)ENC_NW * H * sizeof(float), st, H, I, bt->indptr, bt->indices,
                                  bt->values, o->drop_keep, o->keep_prob, cfg->seed, o->rng_step, gen->p[0], gen->p[4], acts->h1,
                                  acts->row_scale, bt->row_norm2, cfg->item_lo, Ig_of(cfg), pre_only, o->rows_per_step));
}

// the streaming decoder forward (stream_ok): logits of R <= 128 rows over the local slab; stat != NULL: per-group softmax statistics
// as well.  Returns the number of statistic groups.
// (Round 4, built and not run: 64-item tiles with the eight waves as 4 row groups x 2 item halves -- a wave then owns 32 batch rows, so
// every B fragment it reads from LDS feeds two MFMAs: half the LDS bytes per FLOP, the untested suspect for this kernel's 0.46 of the
// HBM rate -- needs two stationary fragment sets = 152 registers per lane: hipcc allocates 256 VGPRs + 776 bytes of scratch per lane for
// it as written, i.e. the variant spills in its inner loop; it would need SGPR-base addressing throughout to fit.)
// (Measured and not kept, round 3: the same loop as TWO independent 4-wave workgroups per CU -- half the batch rows each, 2 x 77 KB of
// LDS, 244 VGPRs, bit-identical outputs -- so that one half's loads overlap the other's MFMAs and stores: 110.4 vs 107.9 us per launch
// at 200 000 items on one box, min 81.8 vs 78.6; whole step 968-970 vs 965-968 us.  The serialisation is not inside the workgroup.)
int launch_dec1_fwd_stream(const ltg_config* cfg, const ltg_gen_state* gen, int R, const ltg_gen_acts* acts, float* stat, hipStream_t st) {
    const int I = cfg->n_items, H = cfg->h_enc;
    // the second form for the HBM-bound slabs (see k_dec1_fwd_stream2); tuning-knob bit 26: the first form at every size, bit 17: the second
    // form from 8 192 items (A/B measurements)
    const int st2_min = (cfg->tuning & (1 << 17)) ? 8192 : ST2_MIN_ITEMS;
    // (the second form addresses the logits AND the shadow rows with 32-bit byte offsets from a uniform base: R * I * 4 and I * ST_KP * 2 must both
    // stay below 2^32 -- 3 532 110 items of one slab for the shadow; beyond either limit the first form, which has none)
    if ((cfg->tuning & (1 << 26)) == 0 && I >= st2_min && (size_t)R * (size_t)I < ((size_t)1 << 30) &&
        (size_t)I * (size_t)(ST_KP * 2) < ((size_t)1 << 32)) {
        const int nt2 = (I + 31) / 32, G2 = nt2 < 256 ? nt2 : 256;
#define LTG_ST2(STATS, NTB) hipLaunchKernelGGL((k_dec1_fwd_stream2<STATS, NTB>), dim3(G2), dim3(ST_NT), (size_t)ST_KS * NTB * 64 * 16, st, R, I, H, acts->h2, gen->wp1t_bf16, gen->p[7], acts->logits, stat)
        if (stat) { if (R <= 112) LTG_ST2(true, 7); else LTG_ST2(true, 8); }
        else { if (R <= 112) LTG_ST2(false, 7); else LTG_ST2(false, 8); }
#undef LTG_ST2
        return G2;
    }
    const int ntiles = (I + ST_BN - 1) / ST_BN, G = ntiles < 256 ? ntiles : 256;
    const size_t lds = (size_t)2 * ST_BN * ST_LDW * 2;
    if (stat) hipLaunchKernelGGL(k_dec1_fwd_stream<true>, dim3(G), dim3(ST_NT), lds, st, R, I, H, acts->h2, gen->wp1t_bf16, gen->p[7], acts->logits, stat);
    else hipLaunchKernelGGL(k_dec1_fwd_stream<false>, dim3(G), dim3(ST_NT), lds, st, R, I, H, acts->h2, gen->wp1t_bf16, gen->p[7], acts->logits, (float*)nullptr);
    return G;
}

// stage 2: (bias + tanh of the all-reduced pre-activation,) enc-1, reparameterisation, dec-0, dec-1 over the local slab
// stat (optional scratch of segpart_floats()): the streaming decoder kernel leaves its per-workgroup softmax statistics there;
// returns the number of workgroups that wrote them (0: the caller reads the logits for the statistics)
int fwd_stage_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_fwd_opts* o,
                   const ltg_gen_acts* acts, int apply_bias_tanh, hipStream_t st, float* stat = nullptr) {
    int stat_groups = 0;
    const int R = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, Z = cfg->z_dim;
    const Probe pr{o->probe, st};
    const bool vz = (Z % 4) == 0 && (cfg->tuning & 8192) == 0;   // 16-B loaders of the middle layers (H % 4 == 0 always)
    if (apply_bias_tanh) {
        const int n = R * H;
        hipLaunchKernelGGL(k_bias_tanh, dim3((n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024), dim3(NT), 0, st, n, H, gen->p[4], acts->h1);
    }
    if (mid_fast(cfg, R)) {
        LTG_PROBED(pr, LTG_K_ENC1, hipLaunchKernelGGL(fk_enc1<false>, grid2(Z, R, 16, 16), dim3(ENC1_NT), 0, st, R, H, Z, acts->h1, gen->p[1], gen->p[5], o->eps, o->is_training,
                                                      cfg->seed, o->rng_step, acts->mulv, acts->z));
        LTG_PROBED(pr, LTG_K_DEC0, hipLaunchKernelGGL(fk_dec0, grid2(H, R, 16, 16), dim3(NT), 0, st, R, H, Z, acts->z, acts->mulv, gen->p[2], gen->p[6],
                                                      acts->kl_rows, acts->h2));
    } else {
    pr.before(LTG_K_ENC1);
    if (vz) hipLaunchKernelGGL((k_dense_fwd<0, true>), grid2(2 * Z, R, 32, 32), dim3(NT), 0, st, R, 2 * Z, H, acts->h1, gen->p[1], gen->p[5], acts->mulv);
    else hipLaunchKernelGGL((k_dense_fwd<0, false>), grid2(2 * Z, R, 32, 32), dim3(NT), 0, st, R, 2 * Z, H, acts->h1, gen->p[1], gen->p[5], acts->mulv);
    pr.after(LTG_K_ENC1);
    hipLaunchKernelGGL(k_reparam, dim3(R), dim3(NT), 0, st, Z, acts->mulv, o->eps, o->is_training, cfg->seed, o->rng_step,
                       acts->z, acts->kl_rows);
    pr.before(LTG_K_DEC0);
    if (vz) hipLaunchKernelGGL((k_dense_fwd<1, true>), grid2(H, R, 32, 32), dim3(NT), 0, st, R, H, Z, acts->z, gen->p[2], gen->p[6], acts->h2);
    else hipLaunchKernelGGL((k_dense_fwd<1, false>), grid2(H, R, 32, 32), dim3(NT), 0, st, R, H, Z, acts->z, gen->p[2], gen->p[6], acts->h2);
    pr.after(LTG_K_DEC0);
    }
    {
        const bool bf = cfg->precision == LTG_PREC_BF16, big = I >= 8192;
        pr.before(LTG_K_DEC1_FWD);
        if (fast_on(cfg) && I <= RD_MAXI && R <= 256) {
            if (bf) hipLaunchKernelGGL(fk_dec1<true>, grid2(I, R, 16, 16), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
            else hipLaunchKernelGGL(fk_dec1<false>, grid2(I, R, 16, 16), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        } else if (stream_ok(cfg, gen, R)) {
            if (stat && (cfg->tuning & (1 << 21)) == 0)   // (tuning-knob bit 21: statistics from a second pass over the logits)
                stat_groups = launch_dec1_fwd_stream(cfg, gen, R, acts, stat, st);
            else
                launch_dec1_fwd_stream(cfg, gen, R, acts, nullptr, st);
        } else if (bf && big) hipLaunchKernelGGL((k_dec1_fwd<true, true>), grid2(I, R, 64, 128), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        else if (bf && (cfg->tuning & 65536) == 0) hipLaunchKernelGGL((k_dec1_fwd<true, false, true>), grid2(I, R, 32, 32), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        else if (bf) hipLaunchKernelGGL((k_dec1_fwd<true, false>), grid2(I, R, 32, 32), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        else if (big) hipLaunchKernelGGL((k_dec1_fwd<false, true>), grid2(I, R, 64, 128), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        else hipLaunchKernelGGL((k_dec1_fwd<false, false>), grid2(I, R, 32, 32), dim3(NT), 0, st, R, I, H, acts->h2, gen->p[3], gen->p[7], acts->logits);
        pr.after(LTG_K_DEC1_FWD);
    }
    return stat_groups;
}

// bytes of the segment-partial scratch for `rows` rows (the first two carve entries of the workspace)
inline size_t segpart_bytes(const ltg_config* cfg, int rows) {
    return align_up((size_t)rows * RP * sizeof(float)) + align_up(segpart_floats(cfg->n_items, rows) * sizeof(float));
}

int vae_forward_impl(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_fwd_opts* o,
                     const ltg_gen_acts* acts, float* probs_out, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
    const int R = bt->n_rows, I = cfg->n_items;
    if (R <= 0) return LTG_OK;
    fwd_stage_enc(cfg, gen, bt, o, acts, 0, st);
    const bool have_scratch = ws && I > 2 * RS_SEG && ws_bytes >= segpart_bytes(cfg, R);
    float* segpart = have_scratch ? reinterpret_cast<float*>((char*)ws + align_up((size_t)R * RP * sizeof(float))) : nullptr;
    const int sg = fwd_stage_rest(cfg, gen, bt, o, acts, 0, st, segpart);
    if (sg > 0) {
        hipLaunchKernelGGL(k_row_stats_merge, dim3(R), dim3(NT), 0, st, R, sg, I, cfg->item_lo, segpart, bt->indptr, bt->indices, bt->values, acts->logits, 0,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (float*)nullptr, acts->lse);
    } else if (have_scratch) {
        // large item slab: one pass over the logits in (segment, row) blocks, then a per-row merge
        const int nseg = (I + RS_SEG - 1) / RS_SEG;
        hipLaunchKernelGGL(k_row_partial_seg, dim3(nseg, R), dim3(NT), 0, st, I, cfg->item_lo, bt->indptr, bt->indices, bt->values, acts->logits,
                           0, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, segpart);
        hipLaunchKernelGGL(k_row_partial_merge, dim3(R), dim3(64), 0, st, R, nseg, segpart, (float*)nullptr, acts->lse);
    } else
    hipLaunchKernelGGL(k_row_lse, dim3(R), dim3(NT), 0, st, I, acts->logits, acts->lse);
    if (probs_out) {
        const int gx = (I + NT - 1) / NT < 64 ? (I + NT - 1) / NT : 64;
        hipLaunchKernelGGL(k_softmax_write, dim3(gx, R), dim3(NT), 0, st, I, acts->logits, acts->lse, probs_out);
    }
    return check_launch();
}

// forward of one or both towers into ws (A1, A3, y, ds, lrow)
// (precision mode, tile size) -> template instance of a discriminator GEMM kernel
inline int d_mode(const ltg_config* cfg) { return cfg->d_precision == LTG_PREC_BF16 ? 1 : (cfg->d_precision == LTG_PREC_FP8 ? 2 : 0); }
// tile of discriminator GEMM kernel `which` (0 l1, 1 l2, 2 bwd1, 3 bwd2): 32 = scalar loaders (latency-bound default
// sizes); 64 / 128 = 16-B vector loaders for a wide discriminator (every dimension a multiple of 4), sized so that each
// launch still fills the 256 CUs
inline int d_tile(const ltg_config* cfg, int which) {
    const bool wide = cfg->d_h0 >= 512 && cfg->d_h1 + cfg->d_h2 >= 512 && cfg->d_h3 >= 128;
    const bool vec = (cfg->d_h0 % 4) == 0 && (cfg->d_h1 % 4) == 0 && (cfg->d_h2 % 4) == 0 && (cfg->d_h3 % 4) == 0;
    const int knob = (cfg->tuning >> 10) & 7;        // tuning: 1 scalar, 2 all 64, 3 all 128
    if (knob == 1) return 32;
    if (!(wide && vec)) {
        // default sizes (100/150/250/300): l2 and backward stage 1 only touch h12 = 400 and h3 = 300 wide rows -> 16-B
        // loaders on the 32 x 32 tiles; l1 / stage 2 index columns of width h1 = 150 (8-B aligned only) -> scalar
        const bool v12 = ((cfg->d_h1 + cfg->d_h2) % 4) == 0 && (cfg->d_h3 % 4) == 0;
        if (which == 1 || which == 2) return v12 ? -32 : 32;
        return ((cfg->d_h0 % 4) == 0 && (cfg->tuning & 32768) == 0) ? -33 : 32;   // embedding rows (operand A) in 16-B pieces
    }
    if (knob == 2) return 64;
    if (knob == 3) return 128;
    if (which == 1) return d_mode(cfg) == 2 ? 64 : -32;   // l2 (N = h3 = 256): too few 64-tiles to fill the chip unless the loads are the bottleneck
    return which == 3 ? 128 : 64;
}
#define LTG_D_DISPATCH3(KERNEL, MODE, TS, V, GRID, ST, ...)                                                          \
    do {                                                                                                             \
        if ((MODE) == 1) hipLaunchKernelGGL((KERNEL<1, TS, V>), GRID, dim3(NT), 0, ST, __VA_ARGS__);                 \
        else if ((MODE) == 2) hipLaunchKernelGGL((KERNEL<2, TS, V>), GRID, dim3(NT), 0, ST, __VA_ARGS__);            \
        else hipLaunchKernelGGL((KERNEL<0, TS, V>), GRID, dim3(NT), 0, ST, __VA_ARGS__);                             \
    } while (0)
// TSV: tile size; -32 = 32 x 32 tiles with 16-B vector loaders on both operands, -33 = on operand A only
#define LTG_D_DISPATCH(KERNEL, MODE, TSV, GRID, ST, ...)                                                             \
    do {                                                                                                             \
        if ((TSV) == 128) LTG_D_DISPATCH3(KERNEL, MODE, 128, 3, GRID, ST, __VA_ARGS__);                              \
        else if ((TSV) == 64) LTG_D_DISPATCH3(KERNEL, MODE, 64, 3, GRID, ST, __VA_ARGS__);                           \
        else if ((TSV) == -32) LTG_D_DISPATCH3(KERNEL, MODE, 32, 3, GRID, ST, __VA_ARGS__);                          \
        else if ((TSV) == -33) LTG_D_DISPATCH3(KERNEL, MODE, 32, 1, GRID, ST, __VA_ARGS__);                          \
        else LTG_D_DISPATCH3(KERNEL, MODE, 32, 0, GRID, ST, __VA_ARGS__);                                            \
    } while (0)

void disc_forward(const ltg_config* cfg, const ltg_disc_state* d, PairView pv, DropView dA, DropView dB, DropView dC,
                  float keep, uint64_t step, const Workspace& w0, bool with_bwd, const ltg_probe* probe, hipStream_t st, float* y_dst = nullptr) {
    Workspace w = w0;
    if (y_dst) w.y = y_dst;   // y straight into the caller's buffer
    const Probe pr{probe, st};
    const int n = pv.nr + pv.nf, h0 = cfg->d_h0, h1 = cfg->d_h1, h2 = cfg->d_h2, h3 = cfg->d_h3, h12 = h1 + h2;
    const int nmax = h1 > h2 ? h1 : h2;
    if (!with_bwd && ft_wsp_capable(cfg) && (cfg->d_arith & 3) != LTG_DARITH_FP32 && (cfg->tuning & ((1 << 23) | (1 << 20))) == 0) {
        // forward only, split arithmetic: ONE kernel from the id lists to y, A1 stays in LDS (ltg_tower.h).  (Tuning-knob bit 20: the three
        // launches below instead.)  The weights are split first: 1 MB at config.ini's sizes, ~3 us, once per call.
        ltg_ft_u32x4* wsp = reinterpret_cast<ltg_ft_u32x4*>(w.wsp);
        const int nwv = (int)ft_wsp_triples(h0, h1, h2, h3);
        hipLaunchKernelGGL(fkt_split_weights, dim3((nwv + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, st, h0, h1, h2, h3, d->p[0], d->p[2], d->p[4], wsp);
        const dim3 g((n + FT_BM - 1) / FT_BM);
        const size_t lds = ft_lds_bytes(h12);
        pr.before(LTG_K_D_L1);
        const bool inj = dA.real || dA.fake || dB.real || dB.fake || dC.real || dC.fake;
#define LTG_FT_LAUNCH(SPLV, INJV)                                                                                                                              \
    hipLaunchKernelGGL((fkt_d_tower<SPLV, 5, INJV>), g, dim3(FT_NT), lds, st, pv, h0, h1, h2, h3, d->emb, wsp, d->p[1], d->p[3], d->p[5], d->p[6], d->p[7], dA, dB, dC, \
                       keep, cfg->seed, step, w.y)
        if ((cfg->d_arith & 3) == LTG_DARITH_BF16X4) { if (inj) LTG_FT_LAUNCH(4, true); else LTG_FT_LAUNCH(4, false); }
        else { if (inj) LTG_FT_LAUNCH(6, true); else LTG_FT_LAUNCH(6, false); }
#undef LTG_FT_LAUNCH
        pr.after(LTG_K_D_L1);
        return;
    }
    if (d_fast(cfg) && !with_bwd && h0 >= 32 && h12 >= 32 && (h0 % 4) == 0 && (h12 % 4) == 0 && (cfg->tuning & (1 << 23)) == 0) {
        // forward only (the fake tower of the G steps, one batch or -- ltg_fake_tower_batched -- 10^5 pair rows): LDS-staged 64 x 64
        // tiles.  The choice depends on the layer sizes only, so the tower inside a step and the batched tower run the same
        // kernels and produce the same bits.  (Tuning-knob bit 23: the register-resident 32 x 32 tiles.)
        LTG_PROBED(pr, LTG_K_D_L1, hipLaunchKernelGGL(fks_d_l1, dim3((h1 + 63) / 64 + (h2 + 63) / 64, (n + 63) / 64), dim3(NT), 0, st, pv, h0, h1, h2, d->emb, d->p[0],
                                                      d->p[1], d->p[2], d->p[3], dA, dB, keep, cfg->seed, step, w.A1));
        LTG_PROBED(pr, LTG_K_D_L2, hipLaunchKernelGGL(fks_d_l2, dim3((h3 + 63) / 64, (n + 63) / 64), dim3(NT), 0, st, n, h12, h3, w.A1, d->p[4], d->p[5], d->p[6], dC, keep,
                                                      cfg->seed, step, w.spart));
        hipLaunchKernelGGL(fk_d_y, dim3((n + NT - 1) / NT), dim3(NT), 0, st, pv, 2 * ((h3 + 63) / 64), w.spart, d->p[7], w.y);
        return;
    }
    if (d_fast(cfg)) {
        // A3, G3 (backward only) and the per-tile partial dot products of the output unit; y only when nothing else follows
        LTG_PROBED(pr, LTG_K_D_L1, LTG_D_SPL_LAUNCH(d_spl(cfg, 0), fk_d_l1, grid2(nmax, n, 32, 32, 2), dim3(NT), 0, st, pv, h0, h1, h2, d->emb, d->p[0], d->p[1], d->p[2],
                                                    d->p[3], dA, dB, keep, cfg->seed, step, w.A1));
        LTG_PROBED(pr, LTG_K_D_L2, LTG_D_SPL_LAUNCH(d_spl(cfg, 1), fk_d_l2, dim3(((h3 + 31) / 32) * ((n + 31) / 32)), dim3(DL2_NT), 0, st, n, h12, h3, w.A1, d->p[4], d->p[5], d->p[6], dC, keep,
                                                    cfg->seed, step, w.A3, with_bwd ? w.G3 : (float*)nullptr, w.spart));
        if (!with_bwd) hipLaunchKernelGGL(fk_d_y, dim3((n + NT - 1) / NT), dim3(NT), 0, st, pv, (h3 + 31) / 32, w.spart, d->p[7], w.y);
        return;
    }
    const int md = d_mode(cfg), ts = d_tile(cfg, 0), ts2 = d_tile(cfg, 1), t1 = ts < 0 ? 32 : ts, t2 = ts2 < 0 ? 32 : ts2;
    if (md == 2 && fast_on(cfg) && d->emb_fp8 && d->w1t_fp8 && d->w2t_fp8 && d->w3t_fp8 && (h0 % 64) == 0 && (h12 % 64) == 0) {
        // operand-format storage: both forward layers read e4m3 bytes (embedding table, transposed weight shadows, A1 in e4m3)
        const bool staged = (h0 % 128) == 0 && (h12 % 128) == 0 && (cfg->tuning & (1 << 24)) == 0;   // LDS-staged tiles (knob bit 24: register-resident)
        if (with_bwd && d_fp8_opfmt(cfg, d)) {
            // the step's own forward: the same products, and every activation the backward multiplies is left behind in e4m3 in the
            // orientation its GEMM contracts over (ltg_fp8bwd.h)
            const int NP = d8_np(n);
            hipLaunchKernelGGL(k8_gather_t, dim3(h0 / 64, NP / 64, 2), dim3(NT), 0, st, pv, h0, NP, d->emb_fp8, w.ET_8);
            // (Round 5, measured and removed: the branch layers' product on 128 x 64 / 128 x 128 tiles -- half the operand bytes through the L1s per
            // output -- ran 38.3 / 72.1 us against the 64 x 64 tiles' 30.1 us on the same box: the loop is bound by the latency of its staged K
            // blocks, which 720 small workgroups hide better than 360 / 180 large ones, not by L2 bandwidth.)
            LTG_PROBED(pr, LTG_K_D_L1, hipLaunchKernelGGL((fk8t_d_l1<64, 64>), dim3((h1 + 63) / 64 + (h2 + 63) / 64, NP / 64), dim3(NT), 0, st, pv, h0, h1, h2, NP,
                                                          d->emb_fp8, d->w1t_fp8, d->p[1], d->w2t_fp8, d->p[3], dA, dB, keep, cfg->seed, step, w.dA1T_16, w.A1_8, w.A1T_8));
            LTG_PROBED(pr, LTG_K_D_L2, hipLaunchKernelGGL((fk8s_d_l2<64, 64>), grid2(h3, n, 64, 64), dim3(NT), 0, st, n, h12, h3, w.A1_8, d->w3t_fp8, d->p[5], dC, keep,
                                                          cfg->seed, step, w.A3));
            hipLaunchKernelGGL(k8_d_out, dim3(NP / 16), dim3(NT), 0, st, pv, h3, NP, w.A3, d->p[6], d->p[7], keep, w.y, w.ds, w.lrow, w.dpre3_8, w.dpre3T_8);
            return;
        }
        if (staged) {
            // 64 x 64 tiles: 696 workgroups of 37 KB LDS, three or four per CU hide each other's load latency (measured at 1 820 pair
            // rows: 30-32 us; 128 x 128 tiles = 180 workgroups, one per CU, 54 us; 128 x 64: 60 us; register-resident block: 52 us)
            LTG_PROBED(pr, LTG_K_D_L1, hipLaunchKernelGGL((fk8s_d_l1<64, 64>), dim3((h1 + 63) / 64 + (h2 + 63) / 64, (n + 63) / 64), dim3(NT), 0, st, pv, h0, h1, h2,
                                                          d->emb_fp8, d->w1t_fp8, d->p[1], d->w2t_fp8, d->p[3], dA, dB, keep, cfg->seed, step, w.A1, w.A1_8));
            LTG_PROBED(pr, LTG_K_D_L2, hipLaunchKernelGGL((fk8s_d_l2<64, 64>), grid2(h3, n, 64, 64), dim3(NT), 0, st, n, h12, h3, w.A1_8, d->w3t_fp8, d->p[5], dC, keep,
                                                          cfg->seed, step, w.A3));
        } else {
        LTG_PROBED(pr, LTG_K_D_L1, hipLaunchKernelGGL(fk8_d_l1, dim3(8 * (((h1 + 63) / 64 + (h2 + 63) / 64 + 7) / 8) * ((n + 63) / 64)), dim3(NT), 0, st, pv, h0, h1, h2, d->emb_fp8, d->w1t_fp8, d->p[1],
                                                      d->w2t_fp8, d->p[3], dA, dB, keep, cfg->seed, step, w.A1, w.A1_8));
        LTG_PROBED(pr, LTG_K_D_L2, hipLaunchKernelGGL(fk8_d_l2, grid2(h3, n, 64, 64), dim3(NT), 0, st, n, h12, h3, w.A1_8, d->w3t_fp8, d->p[5], dC, keep,
                                                      cfg->seed, step, w.A3));
        }
        const dim3 go8((n + NT / 64 - 1) / (NT / 64));
        if (with_bwd) hipLaunchKernelGGL(k_d_out<true>, go8, dim3(NT), 0, st, pv, h3, w.A3, d->p[6], d->p[7], keep, w.y, w.ds, w.lrow, w.dpre3);
        else hipLaunchKernelGGL(k_d_out<false>, go8, dim3(NT), 0, st, pv, h3, w.A3, d->p[6], d->p[7], keep, w.y, w.ds, w.lrow, w.dpre3);
        return;
    }
    LTG_PROBED(pr, LTG_K_D_L1, LTG_D_DISPATCH(k_d_l1, md, ts, grid2(nmax, n, t1, t1, 2), st, pv, h0, h1, h2, d->emb, d->p[0], d->p[1], d->p[2],
                                              d->p[3], dA, dB, keep, cfg->seed, step, w.A1));
    LTG_PROBED(pr, LTG_K_D_L2, LTG_D_DISPATCH(k_d_l2, md, ts2, grid2(h3, n, t2, t2), st, n, h12, h3, w.A1, d->p[4], d->p[5], dC, keep, cfg->seed, step, w.A3));
    const dim3 go((n + NT / 64 - 1) / (NT / 64));
    if (with_bwd) hipLaunchKernelGGL(k_d_out<true>, go, dim3(NT), 0, st, pv, h3, w.A3, d->p[6], d->p[7], keep, w.y, w.ds, w.lrow, w.dpre3);
    else hipLaunchKernelGGL(k_d_out<false>, go, dim3(NT), 0, st, pv, h3, w.A3, d->p[6], d->p[7], keep, w.y, w.ds, w.lrow, w.dpre3);
}

}  // namespace

extern "C" {

int32_t ltg_abi_version(void) { return LTG_ABI_VERSION; }

size_t ltg_workspace_bytes(const ltg_config* cfg, int32_t max_rows, int32_t max_pairs) {
    if (!cfg_ok(cfg) || max_rows < 0 || max_pairs < 0) return 0;
    return carve(cfg, max_rows < 1 ? 1 : max_rows, max_pairs < 1 ? 1 : max_pairs, nullptr).bytes;
}

int ltg_vae_forward(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* batch, const ltg_fwd_opts* opts,
                    const ltg_gen_acts* acts, float* probs_out, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !gen || !batch || !opts || !acts || batch->n_rows < 0) return LTG_EINVAL;
    if (!batch->indptr || !batch->indices || !acts->h1 || !acts->mulv || !acts->z || !acts->h2 || !acts->logits || !acts->lse ||
        !acts->kl_rows || !acts->row_scale)
        return LTG_EINVAL;
    if (opts->rows_per_step < 0 || (opts->rows_per_step > 0 && (opts->is_training != 0.f || opts->drop_keep))) return LTG_EINVAL;
    return vae_forward_impl(cfg, gen, batch, opts, acts, probs_out, (hipStream_t)stream, ws, ws_bytes);
}

size_t ltg_forward_scratch_bytes(const ltg_config* cfg, int32_t max_rows) {
    if (!cfg_ok(cfg) || max_rows <= 0) return 0;
    return segpart_bytes(cfg, max_rows);
}

int ltg_sample_pairs(const ltg_config* cfg, const ltg_sample_inputs* in, const float* logits, const float* lse,
                     int32_t* gen_out, int32_t* pop_out, int32_t* cnt_out, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !in || (!logits && !in->cand_logit) || !lse || !gen_out || !pop_out || !cnt_out) return LTG_EINVAL;
    if (in->n_rows < 0 || in->max_cand < 0) return LTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (in->rows_per_step < 0) return LTG_EINVAL;
    const int groups = in->rows_per_step > 0 ? (in->n_rows + in->rows_per_step - 1) / in->rows_per_step : 1;
    if (hipMemsetAsync(cnt_out, 0, sizeof(int32_t) * (size_t)(groups > 0 ? groups : 1), st) != hipSuccess) return LTG_ELAUNCH;
    if (in->n_rows == 0) return LTG_OK;
    const int max_cand = in->max_cand > 0 ? in->max_cand : 1;
    const size_t lds = (size_t)max_cand * sizeof(float);
    if (lds > 64 * 1024) return LTG_EINVAL;
    hipLaunchKernelGGL(k_sample_pairs, dim3(in->n_rows), dim3(SP_NT), lds, st, Ig_of(cfg), in->cand_ptr, in->cand_idx, in->pop_ptr,
                       in->pop_idx, in->n_sample, in->slot_ptr, in->valid_item, in->u_gumbel, in->u_pick, cfg->seed, in->rng_step,
                       logits, lse, gen_out, pop_out, cnt_out, in->cand_logit, in->rows_per_step);
    return check_launch();
}

// One Adam sweep of the discriminator from `ks` gradient slabs of stride `stride` (lrow: the per-row loss terms of the
// round-1 kernels, else the loss sits in slot P of the slabs)
// one flat float4 sweep (fk_d_adam) when the eight tensors (and moments) lie back to back -- and no operand-format shadows have to follow
// the weights (fp8 mode: k_d_adam rewrites the e4m3 copies of the three matrices it updates)
static bool d_adam_flat(const ltg_config* cfg, const ltg_disc_state* disc, const DLayout& L, int stride, const float* slab, const float* lrow) {
    bool flat = lrow == nullptr && (stride % 4) == 0 && !(cfg->d_precision == LTG_PREC_FP8 && disc->w1t_fp8);
    for (int i = 0; i < 7; ++i) {
        const size_t sz = (size_t)(L.off[i + 1] - L.off[i]);
        flat = flat && disc->p[i + 1] == disc->p[i] + sz && disc->m[i + 1] == disc->m[i] + sz && disc->v[i + 1] == disc->v[i] + sz;
    }
    return flat && ((uintptr_t)disc->p[0] % 16) == 0 && ((uintptr_t)disc->m[0] % 16) == 0 && ((uintptr_t)disc->v[0] % 16) == 0 &&
           ((uintptr_t)slab % 16) == 0;
}
static void d_apply(const ltg_config* cfg, const ltg_disc_state* disc, const DLayout& L, int ks, int stride, const float* slab, int n,
                    const float* lrow, const AdamC& ad, float* loss_out, const Probe& pr, hipStream_t st, const unsigned* poison = nullptr) {
    const int P = L.off[8];
    const bool flat = d_adam_flat(cfg, disc, L, stride, slab, lrow);
    int ga = ((flat ? P / 4 : P) + NT - 1) / NT;
    if (ga > 1024) ga = 1024;
    if (ga < 1) ga = 1;
    if (d_fp8_opfmt(cfg, disc)) {
        // operand-format shadows follow the weights: tile-wise sweep with the transposed e4m3 copies written through LDS
        const int h0 = cfg->d_h0, h1 = cfg->d_h1, h2 = cfg->d_h2, h3 = cfg->d_h3;
        const int nt1 = (h0 / 64) * (h1 / 64), nt2 = (h0 / 64) * (h2 / 64), nt3 = ((h1 + h2) / 64) * (h3 / 64);
        const int nflat = (h1 + h2 + 2 * h3 + 1 + NT - 1) / NT;
        LTG_PROBED(pr, LTG_K_D_ADAM, hipLaunchKernelGGL(k8_d_adam, dim3(nt1 + nt2 + nt3 + nflat + 1), dim3(NT), 0, st, ks, L, stride, h0, h1, h2, h3, nt1, nt2, nt3, slab,
                                                        *disc, ad, n, lrow, loss_out));
        return;
    }
    pr.before(LTG_K_D_ADAM);
    if (flat) hipLaunchKernelGGL(fk_d_adam, dim3(ga + 1), dim3(NT), 0, st, ks, P, stride, slab, disc->p[0], disc->m[0], disc->v[0], ad, loss_out, poison);
    else hipLaunchKernelGGL(k_d_adam, dim3(ga), dim3(NT), 0, st, ks, L, stride, slab, *disc, ad, n, lrow, loss_out);
    pr.after(LTG_K_D_ADAM);
}

// forward + backward of the pair rows in `pv` into gradient slabs; then either the Adam sweep (grad_out == NULL: the
// whole step, train.py:300) or one summed gradient vector in grad_out (this rank's share: ltg_d_grad)
static int d_step_impl(const ltg_config* cfg, const ltg_disc_state* disc, PairView pv, DropView dA, DropView dB, DropView dC,
                       const ltg_d_opts* o, float* grad_out, float* loss_out, const Workspace& w, hipStream_t st) {
    const int n = pv.nr + pv.nf;
    const int h0 = cfg->d_h0, h1 = cfg->d_h1, h2 = cfg->d_h2, h3 = cfg->d_h3, h12 = h1 + h2;
    disc_forward(cfg, disc, pv, dA, dB, dC, o->keep_prob, o->rng_step, w, true, o->probe, st);
    const Probe pr{o->probe, st};
    const AdamC ad = make_adam(cfg, o->adam_t > 0 ? o->adam_t : 1);
    const DLayout L = d_layout(h0, h1, h2, h3);
    const int ks = (n + D_KCHUNK - 1) / D_KCHUNK;
    if (d_fast(cfg)) {
        const int P = L.off[8], SP = d_slab_stride(P), ntile = (h3 + 31) / 32;
        const int nA = (((n + 31) / 32) * ((h12 + 31) / 32) + 7) & ~7;      // padded: job B starts on a multiple of 8 (XCD chunk map)
        const int nB = ks * ((h12 + 1 + 31) / 32) * ((h3 + 31) / 32);
        const int nC = ks * ((h3 + 2 + 31) / 32);
        // Jobs B / C (dw3, db3, dw4, db4, d_loss) need the forward only, not dpre1; job A -> stage 2 -> Adam is the critical chain.  With
        // ltg_d_opts.aux_stream + sync they run on the caller's AUX stream beside job A and stage 2, handed over through two device
        // words like the G step's forks (word 0: the forward is complete, stored by job A's launch when it starts, polled by one wave
        // in front of jobs B / C; word 1: they have ended, polled by an extra block of stage 2 in front of the Adam sweep; word 2: a poll
        // gave up = poison, the sweep then returns at once).  (Measured and not kept, same round: the same jobs riding in stage 2's OWN
        // launch instead of beside job A: 59.5-59.6 -> 61.3-61.6 us per step; profiles/r4_d_step_floor.txt, which also has the step's
        // launch structure: the five grids returning at once take 18 us.)
        // (only with the FLAT tensor layout: the poison word reaches fk_d_adam alone -- separate tensors take k_d_adam, which has no early
        // return, so a direct C-ABI caller with that layout keeps the whole step on one stream)
        const bool fork = o->aux_stream && o->sync && !grad_out && (cfg->tuning & 64) == 0 &&      // (tuning-knob bit 6: no fork)
                          d_adam_flat(cfg, disc, L, SP, w.slab, nullptr);
        const unsigned* poison = fork ? o->sync + 2 : nullptr;
        const int spl1 = d_spl(cfg, 2), spl2 = d_spl(cfg, 3);
        LTG_PROBED(pr, LTG_K_D_BWD1, LTG_D_SPL_LAUNCH(spl1, fk_d_bwd1, dim3(fork ? nA : nA + nB + nC), dim3(NT), 0, st, pv, h12, h3, nA, nB, ntile, L, SP, w.A1, w.A3, w.G3,
                                                      w.spart, disc->p[7], disc->p[4], o->keep_prob, w.dpre1, w.slab,
                                                      fork ? LtgGate{o->sync, o->seq, nullptr, 0} : LTG_NO_GATE));
        if (fork) {
            hipStream_t ax = (hipStream_t)o->aux_stream;
            hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, ax, LtgGate{o->sync, o->seq, o->sync + 2, 0}, LTG_NO_GATE);
            LTG_D_SPL_LAUNCH(spl1, fk_d_bwd1, dim3(nB + nC), dim3(NT), 0, ax, pv, h12, h3, 0, nB, ntile, L, SP, w.A1, w.A3, w.G3, w.spart, disc->p[7], disc->p[4],
                             o->keep_prob, w.dpre1, w.slab, LTG_NO_GATE);
            hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, ax, LtgGate{o->sync + 1, o->seq, nullptr, 0}, LTG_NO_GATE);
        }
        const int n2 = ks * ((h0 + 1 + 15) / 16) * ((h1 + 31) / 32 + (h2 + 31) / 32);
        LTG_PROBED(pr, LTG_K_D_BWD2, LTG_D_SPL_LAUNCH(spl2, fk_d_bwd2, dim3(n2 + (fork ? 1 : 0)), dim3(DB2_NT), 0, st, pv, h0, h1, h2, L, SP, disc->emb, w.dpre1, w.slab,
                                                      fork ? LtgGate{o->sync + 1, o->seq, o->sync + 2, 0} : LTG_NO_GATE));
        if (grad_out) hipLaunchKernelGGL(k_d_grad_sum, dim3(64), dim3(NT), 0, st, ks, P, SP, w.slab, 0, (const float*)nullptr, grad_out);
        else d_apply(cfg, disc, L, ks, SP, w.slab, 0, nullptr, ad, loss_out, pr, st, poison);
        return check_launch();
    }
    if (d_fp8_opfmt(cfg, disc)) {
        // fp8 operands in operand format (ltg_fp8bwd.h): the forward left A1^T, dpre3, dpre3^T and the gathered embeddings^T behind
        const int NP = d8_np(n), P = L.off[8], SP = d_slab_stride(P), ks8 = (NP + D8_KCHUNK - 1) / D8_KCHUNK;
        const int nA = (NP / 64) * ((h12 + 63) / 64);
        const int nB = ks8 * ((h12 + 1 + 63) / 64) * ((h3 + 63) / 64);
        const int nC = ks8 * ((h3 + 1 + 31) / 32);
        LTG_PROBED(pr, LTG_K_D_BWD1, hipLaunchKernelGGL(k8_d_bwd1, dim3(nA + nB + nC), dim3(NT), 0, st, n, NP, h12, h3, nA, nB, L, SP, w.dA1T_16, w.A3, w.ds, w.dpre3_8, w.dpre3T_8,
                                                        w.A1T_8, disc->w3_fp8, o->keep_prob, w.dpre1T_8, w.slab));
        const int n2 = ks8 * ((h0 + 1 + 63) / 64) * (h1 / 64 + h2 / 64);
        LTG_PROBED(pr, LTG_K_D_BWD2, hipLaunchKernelGGL(k8_d_bwd2, dim3(n2), dim3(NT), 0, st, NP, h0, h1, h2, L, SP, w.ET_8, w.dpre1T_8, w.slab));
        if (grad_out) hipLaunchKernelGGL(k_d_grad_sum, dim3(64), dim3(NT), 0, st, ks8, P, SP, w.slab, n, w.lrow, grad_out);
        else d_apply(cfg, disc, L, ks8, SP, w.slab, n, w.lrow, ad, loss_out, pr, st);
        return check_launch();
    }
    // stage 1 (products with the OLD w3) and stage 2 only write gradient slabs; the single Adam sweep runs last
    int ts = d_tile(cfg, 2);
    {   // backward stage 1 with > 1024 32 x 32 tiles is throughput-bound, not latency-bound: 64 x 64 tiles (measured -5 %)
        const long t32 = (long)((n + 31) / 32) * ((h12 + 31) / 32) + (long)ks * ((h12 + 32) / 32) * ((h3 + 31) / 32);
        if (ts == -32 && t32 > 1024 && (cfg->tuning & 4096) == 0) ts = 64;
    }
    const int md = d_mode(cfg), tsb = d_tile(cfg, 3), ta = ts < 0 ? 32 : ts, tb = tsb < 0 ? 32 : tsb;
    auto tiles = [ta](int x) { return (x + ta - 1) / ta; };
    auto tilesb = [tb](int x) { return (x + tb - 1) / tb; };
    const int nA = tiles(n) * tiles(h12);
    const int nB = ks * tiles(h12 + 1) * tiles(h3);
    const int nC = ks * ((h3 + 1 + 31) / 32);
    LTG_PROBED(pr, LTG_K_D_BWD1, LTG_D_DISPATCH(k_d_bwd1, md, ts, dim3(nA + nB + nC), st, n, h12, h3, nA, nB, ks, L, w.A1, w.A3, w.ds, w.dpre3, disc->p[4],
                                                o->keep_prob, w.dpre1, w.slab));
    const int n2 = ks * tilesb(h0 + 1) * (tilesb(h1) + tilesb(h2));
    LTG_PROBED(pr, LTG_K_D_BWD2, LTG_D_DISPATCH(k_d_bwd2, md, tsb, dim3(n2), st, pv, h0, h1, h2, ks, L, disc->emb, w.dpre1, w.slab));
    if (grad_out) hipLaunchKernelGGL(k_d_grad_sum, dim3(64), dim3(NT), 0, st, ks, L.off[8], L.off[8], w.slab, n, w.lrow, grad_out);
    else d_apply(cfg, disc, L, ks, L.off[8], w.slab, n, w.lrow, ad, loss_out, pr, st);
    return check_launch();
}


int ltg_d_step(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* real, const ltg_pairs* fake,
               const ltg_d_opts* o, float* loss_out, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !disc || !real || !fake || !o || !loss_out || !ws || o->adam_t < 1) return LTG_EINVAL;
    const int n = real->n + fake->n;
    if (real->n < 0 || fake->n < 0) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, 1, n) > ws_bytes) return LTG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(loss_out, 0, sizeof(float), st) == hipSuccess ? LTG_OK : LTG_ELAUNCH;
    const Workspace w = carve(cfg, 1, n, (char*)ws);
    PairView pv{real->n, fake->n, real->pop, real->niche, fake->pop, fake->niche};
    DropView dA{o->drop_real[0], o->drop_fake[0], real->n, 0}, dB{o->drop_real[1], o->drop_fake[1], real->n, 0},
        dC{o->drop_real[2], o->drop_fake[2], real->n, 0};
    return d_step_impl(cfg, disc, pv, dA, dB, dC, o, nullptr, loss_out, w, st);
}

size_t ltg_d_grad_floats(const ltg_config* cfg) {
    if (!cfg_ok(cfg)) return 0;
    return (size_t)d_slab_stride(d_layout(cfg->d_h0, cfg->d_h1, cfg->d_h2, cfg->d_h3).off[8]);
}

int ltg_d_grad(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* real, const ltg_pairs* fake, int32_t row_lo,
               int32_t row_hi, const ltg_d_opts* o, float* grad_out, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !disc || !real || !fake || !o || !grad_out || !ws) return LTG_EINVAL;
    const int n = real->n + fake->n;
    if (real->n < 0 || fake->n < 0 || row_lo < 0 || row_hi < row_lo || row_hi > n) return LTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t gf = ltg_d_grad_floats(cfg);
    const int m = row_hi - row_lo;
    if (m == 0) return hipMemsetAsync(grad_out, 0, gf * sizeof(float), st) == hipSuccess ? LTG_OK : LTG_ELAUNCH;
    if (ltg_workspace_bytes(cfg, 1, m) > ws_bytes) return LTG_EWORKSPACE;
    const Workspace w = carve(cfg, 1, m, (char*)ws);
    // the sub-range [row_lo, row_hi) of the logical concatenation real | fake is again a (real, fake) pair of ranges
    const int r0 = row_lo < real->n ? row_lo : real->n, r1 = row_hi < real->n ? row_hi : real->n;
    const int f0 = (row_lo > real->n ? row_lo : real->n) - real->n, f1 = (row_hi > real->n ? row_hi : real->n) - real->n;
    PairView pv{r1 - r0, f1 - f0, real->pop + r0, real->niche + r0, fake->pop + f0, fake->niche + f0};
    const int wd[3] = {cfg->d_h1, cfg->d_h2, cfg->d_h3};
    DropView dv[3];
    for (int i = 0; i < 3; ++i) {
        dv[i].real = o->drop_real[i] ? o->drop_real[i] + (size_t)r0 * wd[i] : nullptr;
        dv[i].fake = o->drop_fake[i] ? o->drop_fake[i] + (size_t)f0 * wd[i] : nullptr;
        dv[i].nr = r1 - r0;
        dv[i].row0 = row_lo;
    }
    return d_step_impl(cfg, disc, pv, dv[0], dv[1], dv[2], o, grad_out, nullptr, w, st);
}

int ltg_d_apply(const ltg_config* cfg, const ltg_disc_state* disc, const float* grad, int32_t adam_t, float* loss_out, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !disc || !grad || !loss_out || adam_t < 1) return LTG_EINVAL;
    const DLayout L = d_layout(cfg->d_h0, cfg->d_h1, cfg->d_h2, cfg->d_h3);
    const Probe pr{nullptr, (hipStream_t)stream};
    d_apply(cfg, disc, L, 1, d_slab_stride(L.off[8]), grad, 0, nullptr, make_adam(cfg, adam_t), loss_out, pr, (hipStream_t)stream);
    return check_launch();
}

// ---- generator step in four stages; between them an item-sharded run exchanges (1) the encoder
// pre-activation [B,H] (all-reduce), (2) the row partials [B,5] (all-gather), (3) dh2 [B,H] (all-reduce).
// ltg_g_step runs the same stages back to back with one "rank".

static void g_row_partial(const ltg_config* cfg, const ltg_batch* bt, const ltg_pairs* fake, const ltg_gen_acts* acts,
                          float* rowpart, hipStream_t st, float* segpart = nullptr, float* lse = nullptr, int stat_groups = 0) {
    const int I = cfg->n_items, B = bt->n_rows;
    if (stat_groups > 0) {   // the decoder kernel left its statistics in segpart
        hipLaunchKernelGGL(k_row_stats_merge, dim3(B), dim3(NT), 0, st, B, stat_groups, I, cfg->item_lo, segpart, bt->indptr, bt->indices, bt->values,
                           acts->logits, fake ? fake->n : 0, fake ? fake->row : nullptr, fake ? fake->niche : nullptr, fake ? fake->pop : nullptr,
                           rowpart, lse);
        return;
    }
    if (segpart && I > 2 * RS_SEG) {
        const int nseg = (I + RS_SEG - 1) / RS_SEG;
        hipLaunchKernelGGL(k_row_partial_seg, dim3(nseg, B), dim3(NT), 0, st, I, cfg->item_lo, bt->indptr, bt->indices, bt->values, acts->logits,
                           fake ? fake->n : 0, fake ? fake->row : nullptr, fake ? fake->niche : nullptr, fake ? fake->pop : nullptr, segpart);
        hipLaunchKernelGGL(k_row_partial_merge, dim3(B), dim3(64), 0, st, B, nseg, segpart, rowpart, lse);
        return;
    }
    hipLaunchKernelGGL(k_row_partial, dim3(bt->n_rows), dim3(NT), 0, st, cfg->n_items, cfg->item_lo, bt->indptr, bt->indices, bt->values,
                       acts->logits, fake ? fake->n : 0, fake ? fake->row : nullptr, fake ? fake->niche : nullptr,
                       fake ? fake->pop : nullptr, rowpart);
}

// dlog as bf16: when BOTH its consumers are the streaming kernels (k_dh2_stream, k_dec1_bwd_adam_stream + ragged tail)
static bool dlog16_ok(const ltg_config* cfg, const ltg_gen_state* gen, int B) {
    return fast_on(cfg) && stream_ok(cfg, gen, B) && dw_stream_ok(cfg->h_enc) && (cfg->tuning & 15) == 0 && (cfg->tuning & (1 << 22)) == 0;
}

static int g_stage_bwd_dec(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* bt,
                           const ltg_pairs* fake, const ltg_g_opts* o, const ltg_gen_acts* acts, const float* rowpart_all,
                           int n_ranks, float* loss_out, const Workspace& w, float* dh2_out, hipStream_t st,
                           bool disc_done = false, const float* h2_for_da2 = nullptr) {
    const int B = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, nf = fake->n;
    const Probe pr{o->probe, st};
    // fake tower forward only (y_data is pruned from the g_trainer fetch, train.py:326); replicated on every rank
    if (nf > 0 && !disc_done) {
        PairView pv{0, nf, nullptr, nullptr, fake->pop, fake->niche};
        DropView dA{nullptr, o->drop_fake[0], 0, 0}, dB{nullptr, o->drop_fake[1], 0, 0}, dC{nullptr, o->drop_fake[2], 0, 0};
        disc_forward(cfg, disc, pv, dA, dB, dC, o->d_keep_prob, o->d_rng_step, w, false, o->probe, st);
    }
    const bool d16 = dlog16_ok(cfg, gen, B);
    if (fast_on(cfg)) {
#define LTG_DLC(D16)                                                                                                                                \
    hipLaunchKernelGGL(k_dlogits_combine<D16>, dim3((I + DL_SEG - 1) / DL_SEG, B), dim3(NT), 0, st, B, I, n_ranks, bt->indptr, bt->indices, bt->values, \
                       acts->logits, rowpart_all, acts->kl_rows, nf > 0 ? w.y : (const float*)nullptr, o->cnt, o->anneal, o->gan_lambda, nf,       \
                       fake->row, fake->niche, fake->pop, w.dlog, acts->lse, w.scal, loss_out, cfg->item_lo)
        if (d16) LTG_DLC(true);
        else LTG_DLC(false);
#undef LTG_DLC
    } else {
    hipLaunchKernelGGL(k_g_combine, dim3(1), dim3(NT), 0, st, B, n_ranks, rowpart_all, nf, acts->kl_rows, nf > 0 ? w.y : nullptr, o->cnt,
                       o->anneal, o->gan_lambda, acts->lse, w.nb, w.Pb, w.scal, loss_out);
    hipLaunchKernelGGL(k_dlogits, dim3((I + DL_SEG - 1) / DL_SEG, B), dim3(NT), 0, st, B, I, bt->indptr, bt->indices, bt->values,
                       acts->logits, acts->lse, w.nb, w.Pb, w.scal, nf, fake->row, fake->niche, fake->pop, w.dlog, cfg->item_lo);
    }
    const bool stream = stream_ok(cfg, gen, B);
    const int kchunk = stream ? dh2_stream_chunk(I) : dh2_kchunk(I);
    const int nsplit = (I + kchunk - 1) / kchunk;
    const bool bf = cfg->precision == LTG_PREC_BF16;
    const bool big = I >= 8192;
    pr.before(LTG_K_DH2);
    if (stream && d16) hipLaunchKernelGGL((k_dh2_stream<true, DH2_NH>), dim3(nsplit, DH2_NH), dim3(ST_NT), (size_t)2 * ST_BN * ST_LDW * 2, st, B, I, H, kchunk, w.dlog, gen->wp1t_bf16, w.part);
    else if (stream) hipLaunchKernelGGL((k_dh2_stream<false, DH2_NH>), dim3(nsplit, DH2_NH), dim3(ST_NT), (size_t)2 * ST_BN * ST_LDW * 2, st, B, I, H, kchunk, w.dlog, gen->wp1t_bf16, w.part);
    else if (bf && big) hipLaunchKernelGGL((k_dh2_partial<true, true>), grid2(H, B, 64, 128, nsplit), dim3(NT), 0, st, B, I, H, kchunk, w.dlog, gen->p[3], w.part);
    else if (bf && (I % 4) == 0 && (cfg->tuning & 65536) == 0) hipLaunchKernelGGL((k_dh2_partial<true, false, true>), grid2(H, B, 32, 32, nsplit), dim3(NT), 0, st, B, I, H, kchunk, w.dlog, gen->p[3], w.part);
    else if (bf) hipLaunchKernelGGL((k_dh2_partial<true, false>), grid2(H, B, 32, 32, nsplit), dim3(NT), 0, st, B, I, H, kchunk, w.dlog, gen->p[3], w.part);
    else if (big) hipLaunchKernelGGL((k_dh2_partial<false, true>), grid2(H, B, 64, 128, nsplit), dim3(NT), 0, st, B, I, H, kchunk, w.dlog, gen->p[3], w.part);
    else hipLaunchKernelGGL((k_dh2_partial<false, false>), grid2(H, B, 32, 32, nsplit), dim3(NT), 0, st, B, I, H, kchunk, w.dlog, gen->p[3], w.part);
    pr.after(LTG_K_DH2);
    {
        const int n = B * H;
        const int gx = (n + NT - 1) / NT < 2048 ? (n + NT - 1) / NT : 2048;
        // slab sum; the single-GPU path folds the tanh derivative in (dh2_out is then already da2)
        hipLaunchKernelGGL(k_da2, dim3(gx), dim3(NT), 0, st, n, nsplit, w.part, h2_for_da2, dh2_out);
    }
    return check_launch();
}

// item -> gradient-row map of the batch: the caller's cache, or rebuilt in the workspace (ltg_batch.slot == NULL)
static const int32_t* g_slot_map(const ltg_config* cfg, const ltg_batch* bt, const Workspace& w, hipStream_t st) {
    if (bt->slot) return bt->slot;
    const int I = cfg->n_items, nu = bt->n_unique;
    hipLaunchKernelGGL(k_fill_i32, dim3((I + NT - 1) / NT < 512 ? (I + NT - 1) / NT : 512), dim3(NT), 0, st, I, -1, w.slotmap);
    if (nu > 0) hipLaunchKernelGGL(k_slot_scatter, dim3((nu + NT - 1) / NT), dim3(NT), 0, st, nu, bt->uptr, bt->csr_pos, bt->indices, w.slotmap);
    return w.slotmap;
}

// sparse gradient rows of W_q0 (+ partial bias rows) into w.gq0
static void g_enc0_grad(const ltg_config* cfg, const ltg_batch* bt, const ltg_g_opts* o, const ltg_gen_acts* acts, const Workspace& w,
                        hipStream_t st, const ltg_gen_state* gen = nullptr, const AdamC* ad = nullptr, bool row_waves = true,
                        const unsigned* poison = nullptr, LtgGate started = LTG_NO_GATE, LtgGate end_wait = LTG_NO_GATE, float* lr_slot = nullptr) {   // gen + ad: fused lazy Adam step
    const int B = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, nu = bt->n_unique;
    const Probe pe{o->probe, st};
    pe.before(LTG_K_ENC0_GRAD);
    if (fast_on(cfg) && row_waves) {   // one wave per row over all columns: a third of the waves (see fk_enc0_grad_rows)
        const int ncb = (H / 4 + 63) / 64;
        const dim3 g((nu + ENC0_BIAS_PARTS + G0_NW - 1) / G0_NW + (end_wait.word ? 1 : 0));
#define LTG_G0_ROWS(N)                                                                                                                                   \
    hipLaunchKernelGGL(fk_enc0_grad_rows<N>, g, dim3(G0_NT), 0, st, B, I, H, nu, bt->uptr, bt->rowidx, bt->csr_pos, bt->indices, bt->values, o->fwd.drop_keep, \
                       o->fwd.keep_prob, cfg->seed, o->fwd.rng_step, acts->row_scale, w.da1, w.gq0, cfg->item_lo, Ig_of(cfg), gen ? *gen : ltg_gen_state{},     \
                       ad ? *ad : AdamC{}, (gen && ad) ? gen->q0_ord + 1 : 0, bt->uitem, poison, started, end_wait, lr_slot)
        if (ncb == 1) LTG_G0_ROWS(1);
        else if (ncb == 2) LTG_G0_ROWS(2);
        else LTG_G0_ROWS(3);
#undef LTG_G0_ROWS
    } else if (fast_on(cfg))
        hipLaunchKernelGGL(fk_enc0_grad, dim3((H / 4 + 63) / 64, (nu + ENC0_BIAS_PARTS + G0_NW - 1) / G0_NW), dim3(G0_NT), 0, st, B, I, H, nu, bt->uptr, bt->rowidx, bt->csr_pos,
                           bt->indices, bt->values, o->fwd.drop_keep, o->fwd.keep_prob, cfg->seed, o->fwd.rng_step, acts->row_scale, w.da1, w.gq0,
                           cfg->item_lo, Ig_of(cfg), gen ? *gen : ltg_gen_state{}, ad ? *ad : AdamC{}, (gen && ad) ? gen->q0_ord + 1 : 0, bt->uitem);
    else
        hipLaunchKernelGGL(k_enc0_grad, dim3(nu + ENC0_BIAS_PARTS), dim3(NT), (size_t)4 * H * sizeof(float), st, B, I, H, nu, bt->uptr, bt->rowidx,
                           bt->csr_pos, bt->indices, bt->values, o->fwd.drop_keep, o->fwd.keep_prob, cfg->seed, o->fwd.rng_step,
                           acts->row_scale, w.da1, w.gq0, cfg->item_lo, Ig_of(cfg));
    pe.after(LTG_K_ENC0_GRAD);
}

// The backward chain dz -> dh1 and the Adam updates of the step as jobs of three launches of fk_g_tail (see there):
//   stage 0: dz tiles + W_p1t (with_dec1: small item slabs; large ones ran the streaming kernel before)
//   stage 1: dh1 tiles + W_p0        stage 2: W_q1, W_q0 (dense product when slot == NULL, else sweep + sparse rows), scalars
static void g_jobs(int stage, const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_g_opts* o, const ltg_gen_acts* acts,
                   const Workspace& w, const AdamC& ad, const int32_t* slot, bool with_dec1, float* loss_out, hipStream_t st,
                   bool no_q0 = false, bool q0_bias = false, const unsigned* poison = nullptr, LtgGate end_wait = LTG_NO_GATE, bool bias_from_da1 = false) {
    const int B = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, Z = cfg->z_dim;
    TailArgs a;
    a.B = B; a.I = I; a.H = H; a.Z = Z; a.nu = bt->n_unique;
    const bool all = stage < 0;     // stage -1: every Adam job in ONE launch (dz / dh1 were launched on their own)
    a.nz = a.nh = 0;
    a.n1 = ((stage == 0 || all) && with_dec1) ? ((I + 31) / 32) * ((H + 1 + LTG_TAIL_BN - 1) / LTG_TAIL_BN) : 0;
    a.n2 = (stage == 1 || all) ? ((Z + 1 + 31) / 32) * ((H + LTG_TAIL_BN - 1) / LTG_TAIL_BN) : 0;
    a.n3 = (stage == 2 || all) ? ((H + 1 + 31) / 32) * ((2 * Z + LTG_TAIL_BN - 1) / LTG_TAIL_BN) : 0;
    a.n4 = 0;
    if ((stage == 2 || all) && !no_q0) {
        if (!slot) a.n4 = ((I + 1 + 31) / 32) * ((H + LTG_TAIL_BN - 1) / LTG_TAIL_BN);
        else {
            const size_t total = (size_t)(I + 1) * (H / 4);
            size_t gx = (total + NT - 1) / NT;
            if (gx > 262144) gx = 262144;
            a.n4 = (int)gx;
        }
    }
    a.q0_bias = 0;
    if ((stage == 2 || all) && q0_bias) {   // lazy Adam clock: fk_enc0_grad updated the item rows, one block finishes the bias row
        a.n4 = bias_from_da1 ? 0 : 1;      // (bias_from_da1: fk_q0_bias_from_da1 follows on the same stream)
        a.q0_bias = bias_from_da1 ? 0 : 1;
    }
    a.n5 = ((stage == 2 || all) && with_dec1) ? 1 : 0;
    a.Wp0 = gen->p[2]; a.Wq1 = gen->p[1]; a.mulv = acts->mulv; a.eps = o->fwd.eps; a.is_training = o->fwd.is_training;
    a.seed = cfg->seed; a.step = o->fwd.rng_step; a.dmlv_out = w.dmlv; a.da1_out = w.da1;
    a.dlog = w.dlog; a.h2 = acts->h2; a.z = acts->z; a.da2 = w.da2; a.h1 = acts->h1; a.dmlv = w.dmlv; a.G = w.gq0;
    a.xd = (slot || q0_bias) ? nullptr : w.xd;
    a.da1 = w.da1;
    a.slot = slot; a.rowout = w.rowout; a.cnt = o->cnt; a.anneal = o->anneal; a.lam = o->gan_lambda;
    a.loss_out = w.scal; a.loss_out2 = loss_out;
    a.poison = poison; a.n_wait = end_wait.word ? 1 : 0; a.end_wait = end_wait;
    const Probe pr{o->probe, st};
    const int kid = stage == 0 ? LTG_K_DZ : (stage == 1 ? LTG_K_DH1 : LTG_K_G_TAIL);
    pr.before(kid);
    const dim3 g(a.nz + a.nh + a.n1 + a.n2 + a.n3 + a.n4 + a.n5 + a.n_wait);
    if (cfg->precision == LTG_PREC_BF16) hipLaunchKernelGGL(fk_g_tail<true>, g, dim3(NT), 0, st, a, *gen, ad);
    else hipLaunchKernelGGL(fk_g_tail<false>, g, dim3(NT), 0, st, a, *gen, ad);
    if ((stage == 2 || all) && q0_bias && bias_from_da1)
        hipLaunchKernelGGL(fk_q0_bias_from_da1, dim3(((H >> 2) + Q0B_COLS - 1) / Q0B_COLS), dim3(NT), 0, st, B, H, w.da1, *gen, ad, poison);
    pr.after(kid);
}

// dz -> dh1 -> (sparse W_q0 gradient) -> Adam updates as one tail launch.
// Adam step gen->q0_ord + 1 of W_q0 / b_q0 on the lazy clock: the batch's rows with their gradient rows (w.gq0), the bias
// row, then the rotating slice of untouched rows
constexpr int G_AUX_SWEEP = 0x40000000;   // library-internal bit of ltg_g_opts.fake_done: ltg_g_step runs the slice on its aux stream
// the rotating slice of G step gen->q0_ord + 1: rows i = ord (mod period) up to `target`
static void q0_slice_sweep(const ltg_config* cfg, const ltg_gen_state* gen, int target, hipStream_t st) {
    const int I = cfg->n_items, P = gen->q0_period, start = (gen->q0_ord + 1) % P;
    if (start < I) hipLaunchKernelGGL(k_q0_sweep, dim3((I - start + P - 1) / P), dim3(Q0_NT), 0, st, I, cfg->h_enc, start, P, target, *gen, make_adam(cfg, 1));
}
static void q0_lazy_update(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_g_opts* o, const Workspace& w, const AdamC& ad,
                           hipStream_t st, bool rows_done = false) {
    const int I = cfg->n_items, H = cfg->h_enc, nu = bt->n_unique, ord = gen->q0_ord + 1;
    // (rows_done: fk_enc0_grad applied the step to the batch's rows and fk_g_tail to the bias row)
    if (!rows_done) hipLaunchKernelGGL(k_q0_step_touched, dim3(nu + 1), dim3(Q0_NT), 0, st, I, H, nu, bt->uptr, bt->csr_pos, bt->indices, w.gq0, ord, *gen, ad);
    if (!(o->fake_done & G_AUX_SWEEP)) q0_slice_sweep(cfg, gen, ord, st);   // (else: ltg_g_step has the slice on its aux stream, up to ord - 1)
}

static void g_chain(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_g_opts* o, const ltg_gen_acts* acts,
                    const Workspace& w, const AdamC& ad, const int32_t* slot, bool with_dec1, float* loss_out, hipStream_t st, bool lazy = false) {
    const int B = bt->n_rows, H = cfg->h_enc, Z = cfg->z_dim;
    const Probe pr{o->probe, st};
    LTG_PROBED(pr, LTG_K_DZ, hipLaunchKernelGGL(fk_dz, grid2(Z, B, 16, 16), dim3(NT), 0, st, B, Z, H, w.da2, gen->p[2], acts->mulv, o->fwd.eps,
                                                o->fwd.is_training, o->anneal, cfg->seed, o->fwd.rng_step, w.dmlv));
    LTG_PROBED(pr, LTG_K_DH1, hipLaunchKernelGGL(fk_dh1, grid2(H, B, 16, 16), dim3(NT), 0, st, B, H, 2 * Z, w.dmlv, gen->p[1], acts->h1, w.da1));
    const bool fused = lazy && fast_on(cfg) && (cfg->tuning & (1 << 25)) == 0;   // Adam on the batch's rows inside the gradient kernel
    const bool row_waves = (cfg->tuning & (1 << 14)) == 0;   // tuning-knob bit 14: the column-blocked shape of the sparse gradient
    if (fused) g_enc0_grad(cfg, bt, o, acts, w, st, gen, &ad, row_waves);
    else if (slot || lazy) g_enc0_grad(cfg, bt, o, acts, w, st, nullptr, nullptr, row_waves);
    const bool own_sweep = (slot && cfg->n_items >= 8192) || lazy;   // HBM-bound sweep: its own launch at full occupancy (measured 490 vs
                                                                     // 525 us at 200 000 items when it rode in the 118-register job kernel)
    g_jobs(-1, cfg, gen, bt, o, acts, w, ad, slot, with_dec1, loss_out, st, own_sweep, fused);
    if (lazy) {
        LTG_PROBED(pr, LTG_K_ENC0_BWD_ADAM, q0_lazy_update(cfg, gen, bt, o, w, ad, st, fused));
    } else if (own_sweep) {
        const int I = cfg->n_items;
        const size_t total = (size_t)(I + 1) * (H / 4);
        size_t gx = (total + NT - 1) / NT;
        if (gx > 262144) gx = 262144;
        LTG_PROBED(pr, LTG_K_ENC0_BWD_ADAM, hipLaunchKernelGGL(k_enc0_bwd_adam, dim3((unsigned)gx), dim3(NT), 0, st, I, H, bt->n_unique, slot, w.gq0, *gen, ad));
    }
}

static int g_stage_bwd_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_g_opts* o,
                            const ltg_gen_acts* acts, const float* dh2, const Workspace& w, hipStream_t st, bool da2_ready = false,
                            bool only_dec1 = false, int dw_groups = 0, LtgH2Done hd = LtgH2Done{nullptr, nullptr, 0u, nullptr}) {
    const int B = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, Z = cfg->z_dim;
    const AdamC ad = make_adam(cfg, o->adam_t);
    const bool bf = cfg->precision == LTG_PREC_BF16;
    const bool big = I >= 8192;
    const bool vz = (Z % 4) == 0 && (cfg->tuning & 8192) == 0;   // 16-B loaders of the middle layers (H % 4 == 0 always)
    // Everything on `st` unless the lazy clock's chain runs beside the weight update (below).  (Measured and removed: small item slabs,
    // the three weight-gradient + Adam kernels on the aux stream beside da2 -> dz -> dh1 -> sweep -- the event pairs cost more than the
    // overlap saves once the chain's kernels take 6-18 us; I = 200 000, the two HBM sweeps side by side: they only contend.)
    hipStream_t aux = (hipStream_t)o->aux_stream;
    hipEvent_t evf = (hipEvent_t)o->ev_fork;
    hipStream_t s_dw = st;
    if (!da2_ready && !only_dec1) {
        const int n = B * H;
        const int gx = (n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024;
        hipLaunchKernelGGL(k_da2, dim3(gx), dim3(NT), 0, st, n, 1, dh2, acts->h2, w.da2);  // da2 = dh2 * (1 - h2^2)
    }
    // persistent workgroups of the streaming decoder weight update (see launch_dw).  Slabs below 65 536 items: 160 -- the rest of the
    // step runs beside the update, and what a kernel of that chain costs there is mostly how many CUs the update leaves it (round 4,
    // same box, per-rank proxy at 25 024 items: 196 workgroups 173.3-174.1 us per step, 160: 170.1-170.5, 128: 184, 96: 205;
    // 20 000 items: 166.3 / 161.6 / 161.3-162.4 / 179; the update alone takes 89 / 92 / 110 / 122 us with 196 / 160 / 128 / 98)
    int dw_gmax = I / 32 < 2048 ? 160 : 224;
    auto launch_dw = [&]() {
        const Probe prs{o->probe, s_dw};
        prs.before(LTG_K_DEC1_BWD_ADAM);
        const int var = (cfg->tuning & 15) > 0 ? (cfg->tuning & 15) - 1 : (big ? 2 : 0);   // tuning: tuning knob (0 = auto)
        if (stream_ok(cfg, gen, B) && dw_stream_ok(H) && (cfg->tuning & 15) == 0) {
            const int ntl = I / 32;
            if (dlog16_ok(cfg, gen, B)) {   // (the producer, g_stage_bwd_dec, stored dlog as bf16 under the same predicate)
                // persistent workgroups: 224 = 28 per XCD (measured 657 us at 200 000 items; 256: 678, 240: 669, 192: 671) -- and 32 CUs
                // stay free for whatever runs beside it.  Tuning-knob bits 27-30 = k: 256 - 8 k instead.
                const int gk = (cfg->tuning >> 27) & 15;
                int gmax = gk ? 256 - 8 * gk : dw_gmax;
                // ... and no more workgroups than the same number of rounds needs (782 tiles of a 25 024-item slab: 4 rounds with
                // 224 or with 196 workgroups -- 60 CUs left to the chain and the collective running beside it)
                if (!gk && ntl > gmax) gmax = (ntl + (ntl + gmax - 1) / gmax - 1) / ((ntl + gmax - 1) / gmax);
                if (dw_groups > 0) gmax = dw_groups;
                hipLaunchKernelGGL(k_dec1_bwd_adam_stream<true>, dim3(ntl < gmax ? ntl : gmax), dim3(ST_NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, hd);
                if (I % 32)   // ragged tail: the generic tile kernel on the last I % 32 item rows
                    hipLaunchKernelGGL((k_dec1_bwd_adam<true, 2, false, true>), grid2(H + 1, I - ntl * 32, 64, 64), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, ntl * 32,
                                       hd.poison);
            } else {
                hipLaunchKernelGGL(k_dec1_bwd_adam_stream<false>, dim3(ntl < 256 ? ntl : 256), dim3(ST_NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad);
                if (I % 32)
                    hipLaunchKernelGGL((k_dec1_bwd_adam<true, 2>), grid2(H + 1, I - ntl * 32, 64, 64), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, ntl * 32);
            }
        } else if (!bf) {
            if (var == 0) hipLaunchKernelGGL((k_dec1_bwd_adam<false, 0>), grid2(H + 1, I, 32, 32), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
            else hipLaunchKernelGGL((k_dec1_bwd_adam<false, 2>), grid2(H + 1, I, 64, 64), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        } else if (var == 0 && (I % 4) == 0 && (cfg->tuning & 65536) == 0) hipLaunchKernelGGL((k_dec1_bwd_adam<true, 0, true>), grid2(H + 1, I, 32, 32), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        else if (var == 0) hipLaunchKernelGGL((k_dec1_bwd_adam<true, 0>), grid2(H + 1, I, 32, 32), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        else if (var == 1) hipLaunchKernelGGL((k_dec1_bwd_adam<true, 1>), grid2(H + 1, I, 128, 64), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        else if (var == 2) hipLaunchKernelGGL((k_dec1_bwd_adam<true, 2>), grid2(H + 1, I, 64, 64), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        else hipLaunchKernelGGL((k_dec1_bwd_adam<true, 3>), grid2(H + 1, I, 128, 32), dim3(NT), 0, s_dw, B, I, H, w.dlog, acts->h2, *gen, ad, 0);
        prs.after(LTG_K_DEC1_BWD_ADAM);
    };
    if ((o->fake_done & G_AUX_SWEEP) && q0_lazy(cfg, gen) && mid_fast(cfg, B) && !only_dec1) {
        // ltg_g_step, large item slab, lazy Adam clock of W_q0.  The decoder weight update (HBM-bound, the largest kernel of the
        // step) needs only dlog and h2; everything else that is left -- dz -> dh1 -> sparse W_q0 gradient -> the other Adam
        // updates -> the clock's step and its rotating slice -- needs only da2.  The two run side by side: the weight update on
        // `st` with 224 of its 256 persistent workgroups (28 per XCD; measured FASTER than 256: 657 vs 678 us at 200 000 items),
        // the chain on the aux stream on the CUs that leaves free.  (Round 1 found no overlap here: the weight update then held
        // every CU.)  Joined by ev_sweep at the end of ltg_g_step.
        (void)hipEventRecord(evf, st);
        (void)hipStreamWaitEvent(aux, evf, 0);
        // 64 CUs to the chain when the update has many rounds anyway: with Adam moments in every row of W_q0 the clock's deferred
        // arithmetic makes the chain the longer side on 32 CUs (same box, 200 000 items, warm moments: 224 -> 673 ms per 640
        // steps, 208 -> 666, 192 -> 647, 176 -> 650, 160 -> 659; cold moments 629 vs 632)
        if (I / 32 >= 2048) dw_gmax = 192;
        if (!o->dec1_done) launch_dw();
        ltg_g_opts oc = *o;
        oc.fake_done &= ~G_AUX_SWEEP;   // the slice runs in the chain's own stream order, behind the clock's step
        g_chain(cfg, gen, bt, &oc, acts, w, ad, nullptr, false, nullptr, aux, true);
        (void)hipEventRecord((hipEvent_t)o->ev_sweep, aux);
        return check_launch();
    }
    if ((o->fake_done & G_AUX_SWEEP) && q0_lazy(cfg, gen)) {
        // (batches the middle-layer fast path does not serve) only the rotating slice on the aux stream, eligible together with the decoder weight update
        (void)hipEventRecord(evf, st);
        (void)hipStreamWaitEvent(aux, evf, 0);
        if (!o->dec1_done) launch_dw();
        q0_slice_sweep(cfg, gen, gen->q0_ord, aux);   // the batch's rows are at q0_ord already (q0_touch): skipped
        (void)hipEventRecord((hipEvent_t)o->ev_sweep, aux);
    } else if (!o->dec1_done) launch_dw();
    if (only_dec1) return check_launch();
    if (mid_fast(cfg, B)) {
        // dz -> dh1 -> sparse W_q0 gradient, then every remaining Adam update (W_p0, W_q1, W_q0 + biases) in ONE launch
        const bool lazy = q0_lazy(cfg, gen);   // no item -> gradient-row map needed: the update walks the batch's distinct items
        const int32_t* slot = lazy ? nullptr : g_slot_map(cfg, bt, w, st);
        g_chain(cfg, gen, bt, o, acts, w, ad, slot, false, nullptr, st, lazy);
        return check_launch();
    }
    const Probe pc{o->probe, st};
    pc.before(LTG_K_DZ);
#define LTG_V2(KERNEL, ...)                                       \
    do {                                                          \
        if (vz) hipLaunchKernelGGL(KERNEL<true>, __VA_ARGS__);    \
        else hipLaunchKernelGGL(KERNEL<false>, __VA_ARGS__);      \
    } while (0)
    LTG_V2(k_dz, grid2(Z, B, 32, 32), dim3(NT), 0, st, B, Z, H, w.da2, gen->p[2], acts->mulv, o->fwd.eps, o->fwd.is_training,
           o->anneal, cfg->seed, o->fwd.rng_step, w.dmlv);
    pc.after(LTG_K_DZ);
    pc.before(LTG_K_WGRAD_P0);
    LTG_V2(k_wgrad_adam, grid2(H, Z + 1, 32, 32), dim3(NT), 0, st, B, Z, H, acts->z, w.da2, gen->p[2], gen->m[2], gen->v[2],
           gen->p[6], gen->m[6], gen->v[6], ad);
    pc.after(LTG_K_WGRAD_P0);
    pc.before(LTG_K_DH1);
    LTG_V2(k_dh1, grid2(H, B, 32, 32), dim3(NT), 0, st, B, H, 2 * Z, w.dmlv, gen->p[1], acts->h1, w.da1);
    pc.after(LTG_K_DH1);
    pc.before(LTG_K_WGRAD_Q1);
    LTG_V2(k_wgrad_adam, grid2(2 * Z, H + 1, 32, 32), dim3(NT), 0, st, B, H, 2 * Z, acts->h1, w.dmlv, gen->p[1], gen->m[1],
           gen->v[1], gen->p[5], gen->m[5], gen->v[5], ad);
#undef LTG_V2
    pc.after(LTG_K_WGRAD_Q1);
    const int nu = bt->n_unique;
    const Probe pe{o->probe, st};
    pe.before(LTG_K_ENC0_BWD_ADAM);
    hipLaunchKernelGGL(k_enc0_grad, dim3(nu + ENC0_BIAS_PARTS), dim3(NT), (size_t)4 * H * sizeof(float), st, B, I, H, nu, bt->uptr, bt->rowidx,
                       bt->csr_pos, bt->indices, bt->values, o->fwd.drop_keep, o->fwd.keep_prob, cfg->seed, o->fwd.rng_step,
                       acts->row_scale, w.da1, w.gq0, cfg->item_lo, Ig_of(cfg));
    if (q0_lazy(cfg, gen)) q0_lazy_update(cfg, gen, bt, o, w, ad, st);
    else {
        const size_t total = (size_t)(I + 1) * (H / 4);
        size_t gx = (total + NT - 1) / NT;
        if (gx > 262144) gx = 262144;
        const int32_t* slot = bt->slot;
        if (!slot) {   // no per-batch cache (n_batches x I ints at full scale): build the map of this batch in the workspace
            hipLaunchKernelGGL(k_fill_i32, dim3((I + NT - 1) / NT < 512 ? (I + NT - 1) / NT : 512), dim3(NT), 0, st, I, -1, w.slotmap);
            if (nu > 0) hipLaunchKernelGGL(k_slot_scatter, dim3((nu + NT - 1) / NT), dim3(NT), 0, st, nu, bt->uptr, bt->csr_pos, bt->indices, w.slotmap);
            slot = w.slotmap;
        }
        hipLaunchKernelGGL(k_enc0_bwd_adam, dim3((unsigned)gx), dim3(NT), 0, st, I, H, nu, slot, w.gq0, *gen, ad);
    }
    pe.after(LTG_K_ENC0_BWD_ADAM);
    return check_launch();
}

static bool g_args_ok(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_gen_acts* acts) {
    return cfg_ok(cfg) && gen && bt && acts && bt->n_rows > 0 && bt->indptr && bt->indices;
}

int ltg_g_step(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* bt,
               const ltg_pairs* fake, const ltg_g_opts* o, const ltg_gen_acts* acts, float* loss_out, void* ws,
               size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !gen || !disc || !bt || !fake || !o || !acts || !loss_out || !ws || o->adam_t < 1) return LTG_EINVAL;
    if (!bt->uptr || !bt->rowidx || !bt->csr_pos || !o->cnt || !fake->row || bt->n_rows <= 0 || fake->n < 0) return LTG_EINVAL;
    if (bt->n_unique < 0 || (size_t)bt->n_unique > gq0_rows(cfg, bt->n_rows)) return LTG_EINVAL;
    const int B = bt->n_rows, nf = fake->n;
    if (ltg_workspace_bytes(cfg, B, nf) > ws_bytes) return LTG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Workspace w = carve(cfg, B, nf, (char*)ws);
    const bool have_y = o->y_pre != nullptr;   // y_generated of this batch came from ltg_fake_tower_batched: no tower in this step
    if (have_y) w.y = const_cast<float*>(o->y_pre);
    // lazy Adam clock of W_q0: the batch's rows up to date first; its rotating slice (rows NOT of this batch: arithmetic-bound,
    // 48 registers -- it fits beside the 2 x 232-register waves of the HBM-bound decoder kernels) then runs on the aux stream
    q0_touch(cfg, gen, bt, st);
    const bool aux_sweep = q0_lazy(cfg, gen) && o->aux_stream && o->ev_fork && o->ev_sweep && (cfg->tuning & 512) == 0;
    // fork: the fake tower (independent of the generator forward) runs on the caller's aux stream
    const bool fork = o->aux_stream && o->ev_fork && o->ev_join && nf > 0 && (cfg->tuning & 512) == 0 && !have_y;
    if (fork || aux_sweep) {
        hipStream_t aux = (hipStream_t)o->aux_stream;
        if (hipEventRecord((hipEvent_t)o->ev_fork, st) != hipSuccess || hipStreamWaitEvent(aux, (hipEvent_t)o->ev_fork, 0) != hipSuccess)
            return LTG_ELAUNCH;
        if (fork) {
            PairView pv{0, nf, nullptr, nullptr, fake->pop, fake->niche};
            DropView dA{nullptr, o->drop_fake[0], 0, 0}, dB{nullptr, o->drop_fake[1], 0, 0}, dC{nullptr, o->drop_fake[2], 0, 0};
            disc_forward(cfg, disc, pv, dA, dB, dC, o->d_keep_prob, o->d_rng_step, w, false, nullptr, aux);
            if (hipEventRecord((hipEvent_t)o->ev_join, aux) != hipSuccess) return LTG_ELAUNCH;
        }
    }
    ltg_g_opts o_local = *o;
    o_local.fake_done = (o->fake_done & 1) | (aux_sweep ? G_AUX_SWEEP : 0);
    o = &o_local;
    const bool small = small_fast(cfg, B);
    fwd_stage_enc(cfg, gen, bt, &o->fwd, acts, 0, st, small ? w.xd : nullptr, true);
    const int sgroups = fwd_stage_rest(cfg, gen, bt, &o->fwd, acts, 0, st, small ? nullptr : w.segpart);
    if (small) {
        // small item slab: a row's softmax statistics, loss terms and dlogits need no other row -> one launch per stage,
        // nine launches per step: enc0, enc1, dec0, dec1 | row softmax + dlogits, dh2, dz, dh1, Adam tail (dW_q0 = xd^T . da1 dense)
        const int I = cfg->n_items, H = cfg->h_enc;
        const Probe pr{o->probe, st};
        if (nf > 0 && !fork && !have_y) {
            PairView pv{0, nf, nullptr, nullptr, fake->pop, fake->niche};
            DropView dA{nullptr, o->drop_fake[0], 0, 0}, dB{nullptr, o->drop_fake[1], 0, 0}, dC{nullptr, o->drop_fake[2], 0, 0};
            disc_forward(cfg, disc, pv, dA, dB, dC, o->d_keep_prob, o->d_rng_step, w, false, o->probe, st);
        }
        if (fork && hipStreamWaitEvent(st, (hipEvent_t)o->ev_join, 0) != hipSuccess) return LTG_ELAUNCH;
        LTG_PROBED(pr, LTG_K_ROW_DLOGITS, hipLaunchKernelGGL(fk_row_dlogits, dim3(B), dim3(NT), 0, st, B, I, bt->indptr, bt->indices, bt->values, acts->logits,
                                                             acts->kl_rows, w.y, nf, o->cnt, o->gan_lambda, fake->row, fake->niche, fake->pop, w.dlog,
                                                             acts->lse, w.rowout));
        pr.before(LTG_K_DH2);
        if (cfg->precision == LTG_PREC_BF16) hipLaunchKernelGGL(fk_dh2<true>, grid2(H, B, 16, 16), dim3(DH2_NT), 0, st, B, I, H, w.dlog, gen->p[3], acts->h2, w.da2);
        else hipLaunchKernelGGL(fk_dh2<false>, grid2(H, B, 16, 16), dim3(DH2_NT), 0, st, B, I, H, w.dlog, gen->p[3], acts->h2, w.da2);
        pr.after(LTG_K_DH2);
        g_chain(cfg, gen, bt, o, acts, w, make_adam(cfg, o->adam_t), nullptr, true, loss_out, st);
        return check_launch();
    }
    g_row_partial(cfg, bt, fake, acts, w.rowpart, st, w.segpart, nullptr, sgroups);
    if (fork && hipStreamWaitEvent(st, (hipEvent_t)o->ev_join, 0) != hipSuccess) return LTG_ELAUNCH;
    // single GPU: the slab sum writes da2 directly (one launch less than the sharded stage pair)
    int rc = g_stage_bwd_dec(cfg, gen, disc, bt, fake, o, acts, w.rowpart, 1, loss_out, w, w.da2, st, fork || have_y, acts->h2);
    if (rc != LTG_OK) return rc;
    rc = g_stage_bwd_rest(cfg, gen, bt, o, acts, w.da2, w, st, true);
    if (aux_sweep && hipStreamWaitEvent(st, (hipEvent_t)o->ev_sweep, 0) != hipSuccess) return LTG_ELAUNCH;   // join
    return rc;
}

/* ---- the same step cut at its three exchange points (item-sharded multi-GPU; include/ltg.h) ---- */
int ltg_g_fwd_enc(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_fwd_opts* opts,
                  const ltg_gen_acts* acts, ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !opts) return LTG_EINVAL;
    fwd_stage_enc(cfg, gen, bt, opts, acts, 1, (hipStream_t)stream);
    return check_launch();
}

int ltg_g_fwd_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_pairs* fake,
                   const ltg_fwd_opts* opts, const ltg_gen_acts* acts, float* rowpart_out, void* ws, size_t ws_bytes,
                   ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !opts || !rowpart_out) return LTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* segpart = (ws && ws_bytes >= segpart_bytes(cfg, bt->n_rows))
                         ? reinterpret_cast<float*>((char*)ws + align_up((size_t)bt->n_rows * RP * sizeof(float))) : nullptr;
    const int sgroups = fwd_stage_rest(cfg, gen, bt, opts, acts, 1, st, segpart);
    g_row_partial(cfg, bt, (fake && fake->n > 0) ? fake : nullptr, acts, rowpart_out, st, segpart, nullptr, sgroups);
    return check_launch();
}

int ltg_rowstats_combine(const ltg_config* cfg, const float* rowpart_all, int32_t n_ranks, int32_t n_rows, float* lse_out,
                         void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !rowpart_all || !lse_out || n_ranks < 1 || n_rows < 1 || !ws) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, n_rows, 1) > ws_bytes) return LTG_EWORKSPACE;
    const Workspace w = carve(cfg, n_rows, 1, (char*)ws);
    hipLaunchKernelGGL(k_g_combine, dim3(1), dim3(NT), 0, (hipStream_t)stream, n_rows, n_ranks, rowpart_all, 0, (const float*)nullptr,
                       (const float*)nullptr, (const int32_t*)nullptr, 0.f, 0.f, lse_out, w.nb, w.Pb, (float*)nullptr, (float*)nullptr);
    return check_launch();
}

int ltg_g_bwd_dec(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* bt,
                  const ltg_pairs* fake, const ltg_g_opts* o, const ltg_gen_acts* acts, const float* rowpart_all,
                  int32_t n_ranks, float* loss_out, float* dh2_out, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !disc || !fake || !o || !rowpart_all || n_ranks < 1 || !loss_out || !dh2_out || !ws) return LTG_EINVAL;
    if (!o->cnt || !fake->row || fake->n < 0) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, bt->n_rows, fake->n) > ws_bytes) return LTG_EWORKSPACE;
    Workspace w = carve(cfg, bt->n_rows, fake->n, (char*)ws);
    if (o->y_pre) w.y = const_cast<float*>(o->y_pre);   // y_generated from ltg_fake_tower_batched
    return g_stage_bwd_dec(cfg, gen, disc, bt, fake, o, acts, rowpart_all, n_ranks, loss_out, w, dh2_out, (hipStream_t)stream,
                           (o->fake_done & 1) != 0 || o->y_pre != nullptr);
}

int ltg_g_fake_tower(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* fake, const ltg_g_opts* o, int32_t n_rows, void* ws,
                     size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !disc || !fake || !o || !ws || n_rows <= 0 || fake->n < 0) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, n_rows, fake->n) > ws_bytes) return LTG_EWORKSPACE;
    if (fake->n == 0) return LTG_OK;
    const Workspace w = carve(cfg, n_rows, fake->n, (char*)ws);   // same carve as ltg_g_bwd_dec: y lives there
    PairView pv{0, fake->n, nullptr, nullptr, fake->pop, fake->niche};
    DropView dA{nullptr, o->drop_fake[0], 0, 0}, dB{nullptr, o->drop_fake[1], 0, 0}, dC{nullptr, o->drop_fake[2], 0, 0};
    disc_forward(cfg, disc, pv, dA, dB, dC, o->d_keep_prob, o->d_rng_step, w, false, nullptr, (hipStream_t)stream);
    return check_launch();
}

int ltg_fake_tower_batched(const ltg_config* cfg, const ltg_disc_state* disc, const ltg_pairs* fake, const int32_t* seg_of,
                           const int32_t* seg_row0, const uint64_t* seg_step, float d_keep_prob, float* y_out, void* ws, size_t ws_bytes,
                           ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !disc || !fake || !seg_of || !seg_row0 || !seg_step || !y_out || !ws || fake->n < 0 || !fake->pop || !fake->niche) return LTG_EINVAL;
    if (fake->n == 0) return LTG_OK;
    if (ltg_workspace_bytes(cfg, 1, fake->n) > ws_bytes) return LTG_EWORKSPACE;
    const Workspace w = carve(cfg, 1, fake->n, (char*)ws);
    PairView pv{0, fake->n, nullptr, nullptr, fake->pop, fake->niche};
    DropView dv{nullptr, nullptr, 0, 0};
    dv.seg_of = seg_of;
    dv.seg_row0 = seg_row0;
    dv.seg_step = seg_step;
    disc_forward(cfg, disc, pv, dv, dv, dv, d_keep_prob, 0, w, false, nullptr, (hipStream_t)stream, y_out);
    return check_launch();
}

int ltg_g_bwd_rest(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_pairs* fake,
                   const ltg_g_opts* o, const ltg_gen_acts* acts, const float* dh2, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !fake || !o || !dh2 || !ws || o->adam_t < 1) return LTG_EINVAL;
    if (!bt->uptr || !bt->rowidx || !bt->csr_pos) return LTG_EINVAL;
    if (bt->n_unique < 0 || (size_t)bt->n_unique > gq0_rows(cfg, bt->n_rows)) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, bt->n_rows, fake->n) > ws_bytes) return LTG_EWORKSPACE;
    const Workspace w = carve(cfg, bt->n_rows, fake->n, (char*)ws);   // same carve as ltg_g_bwd_dec: dlog lives there
    return g_stage_bwd_rest(cfg, gen, bt, o, acts, dh2, w, (hipStream_t)stream);
}

int ltg_g_bwd_dec1(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_pairs* fake,
                   const ltg_g_opts* o, const ltg_gen_acts* acts, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !fake || !o || !ws || o->adam_t < 1 || o->dec1_done) return LTG_EINVAL;
    if (ltg_workspace_bytes(cfg, bt->n_rows, fake->n) > ws_bytes) return LTG_EWORKSPACE;
    const Workspace w = carve(cfg, bt->n_rows, fake->n, (char*)ws);   // same carve as ltg_g_bwd_dec: dlog lives there
    return g_stage_bwd_rest(cfg, gen, bt, o, acts, nullptr, w, (hipStream_t)stream, true, true);
}

/* ---- the item-sharded step as ONE call, exchanges in-stream, weight update and clock slice beside the next step (include/ltg.h) ---- */
int ltg_g_step_sharded_ok(const ltg_config* cfg, const ltg_gen_state* gen, int32_t n_rows) {
    if (!cfg_ok(cfg) || !gen || n_rows <= 0) return 0;
    return (fast_on(cfg) && mid_fast(cfg, n_rows) && stream_ok(cfg, gen, n_rows) && dw_stream_ok(cfg->h_enc) && dlog16_ok(cfg, gen, n_rows) &&
            q0_lazy(cfg, gen)) ? 1 : 0;
}

// the catch-up ahead (include/ltg.h): device words with the slice on the side stream, marks, batches with their distinct-item lists
static bool q0_ahead_capable(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_pipe* pp) {
    if (!cfg || !gen || !bt || !pp || !pp->q0_mark || !pp->sync || !bt->uitem || !bt->uptr || !bt->csr_pos) return false;
    const int fl = pp->flags;
    if (fl & (LTG_PIPE_NO_DEC1_FORK | LTG_PIPE_NO_SLICE_FORK | LTG_PIPE_EVENTS | LTG_PIPE_SLICE_IN_TOUCH)) return false;
    return q0_lazy(cfg, gen) && (cfg->h_enc >> 2) <= Q0_NT;
}
// the pipe's hand-overs are device words (not events, not program order)
static bool pipe_gates(const ltg_pipe* pp) { return pp->sync && (pp->flags & (LTG_PIPE_NO_DEC1_FORK | LTG_PIPE_EVENTS)) == 0; }
// the weight update writes the pipe's second shadow buffer
static bool shadow_pingpong(const ltg_gen_state* gen, const ltg_pipe* pp) {
    return pipe_gates(pp) && pp->shadow_out && gen->wp1t_bf16 && pp->shadow_out != gen->wp1t_bf16;
}
int ltg_g_step_sharded_plan(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_batch* bt, const ltg_pipe* pp) {
    if (!cfg_ok(cfg) || !gen || !bt || !pp || !ltg_g_step_sharded_ok(cfg, gen, bt->n_rows)) return 0;
    return (q0_ahead_capable(cfg, gen, bt, pp) ? LTG_PLAN_AHEAD : 0) | (shadow_pingpong(gen, pp) ? LTG_PLAN_SHADOW : 0);
}

int ltg_g_pipe_probe(const ltg_pipe* pipe, ltg_stream stream) {
    clear_errors();
    if (!pipe || !pipe->sync || !pipe->side_stream) return LTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // a waiter on the side stream FIRST, then its producer on `stream`: with one hardware queue under both, the producer cannot start
    // before the waiter has given up (5 ms).  The same for the tail stream when the pipe has one.
    // Pairs (waiter's stream, setter's stream): the caller's stream against each side stream, and -- when the pipe has both -- the tail
    // stream against the side stream: they never wait for each other, but in one queue the Adam tail would sit behind the ~100-us weight
    // update and the next call's enc-0, which polls for the tail's end, with it (seen in round 4: a process with many streams alive put
    // both onto one hardware queue: 180 instead of 147 us per step at 20 000 items).
    for (int which = 0; which < 3; ++which) {
        hipStream_t sd = (hipStream_t)(which == 0 ? pipe->side_stream : pipe->tail_stream);
        hipStream_t sp = which == 2 ? (hipStream_t)pipe->side_stream : st;
        if (!sd || (which == 2 && !pipe->side_stream)) continue;     // (`stream` itself may be the null stream: handle 0)
        unsigned zero[2] = {0u, 0u}, got[2] = {0u, 0u};
        if (hipStreamSynchronize(sd) != hipSuccess || hipStreamSynchronize(sp) != hipSuccess) return LTG_ELAUNCH;
        if (hipMemcpy(pipe->sync + 3, zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) return LTG_ELAUNCH;
        hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, sd, LtgGate{pipe->sync + 3, 1u, pipe->sync + 4, 5});
        hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, sp, LtgGate{pipe->sync + 3, 1u, nullptr, 0});
        if (hipStreamSynchronize(sd) != hipSuccess || hipStreamSynchronize(sp) != hipSuccess) return LTG_ELAUNCH;
        if (hipMemcpy(got, pipe->sync + 3, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess) return LTG_ELAUNCH;
        if (check_launch() != LTG_OK) return LTG_ELAUNCH;
        if (!(got[0] == 1u && got[1] == 0u)) return 0;
    }
    return 1;
}

int ltg_g_pipe_join(const ltg_pipe* pipe, ltg_stream stream) {
    clear_errors();
    if (!pipe || !pipe->ev_dec1 || !pipe->side_stream) return LTG_EINVAL;
    if (pipe->sync && (pipe->flags & LTG_PIPE_EVENTS) == 0) {   // gates: nothing was recorded per call -- everything on the side stream so far
        if (hipEventRecord((hipEvent_t)pipe->ev_dec1, (hipStream_t)pipe->side_stream) != hipSuccess) return LTG_ELAUNCH;
    }
    if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)pipe->ev_dec1, 0) != hipSuccess) return LTG_ELAUNCH;
    if (pipe->tail_stream && pipe->ev_tail) {   // the Adam tail of the last call (its own stream in the device-word mode)
        if (hipEventRecord((hipEvent_t)pipe->ev_tail, (hipStream_t)pipe->tail_stream) != hipSuccess) return LTG_ELAUNCH;
        if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)pipe->ev_tail, 0) != hipSuccess) return LTG_ELAUNCH;
    }
    return LTG_OK;
}

int ltg_g_step_sharded(const ltg_config* cfg, const ltg_gen_state* gen, const ltg_disc_state* disc, const ltg_batch* bt,
                       const ltg_pairs* fake, const ltg_g_opts* o, const ltg_gen_acts* acts, const ltg_comm* comm,
                       const ltg_pipe* pp, float* loss_out, void* ws, size_t ws_bytes, ltg_stream stream) {
    clear_errors();
    if (!g_args_ok(cfg, gen, bt, acts) || !disc || !fake || !o || !pp || !loss_out || !ws || o->adam_t < 1) return LTG_EINVAL;
    if (!bt->uptr || !bt->rowidx || !bt->csr_pos || !o->cnt || !fake->row || fake->n < 0) return LTG_EINVAL;
    if (bt->n_unique < 0 || (size_t)bt->n_unique > gq0_rows(cfg, bt->n_rows)) return LTG_EINVAL;
    if (!ltg_g_step_sharded_ok(cfg, gen, bt->n_rows)) return LTG_EINVAL;
    if (!pp->side_stream || !pp->ev_fork || !pp->ev_dec1 || !pp->h1pre || !pp->rowpart_all || !pp->dh2) return LTG_EINVAL;
    const int R = comm ? comm->n_ranks : 1, rank = comm ? comm->rank : 0;
    if (R < 1 || rank < 0 || rank >= R || (comm && (!comm->all_reduce || !comm->all_gather))) return LTG_EINVAL;
    const int B = bt->n_rows, I = cfg->n_items, H = cfg->h_enc, Z = cfg->z_dim, nf = fake->n;
    if (ltg_workspace_bytes(cfg, B, nf) > ws_bytes) return LTG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream, sd = (hipStream_t)pp->side_stream;
    hipEvent_t ev_fork = (hipEvent_t)pp->ev_fork, ev_dec1 = (hipEvent_t)pp->ev_dec1;
    const bool fork_dec1 = (pp->flags & LTG_PIPE_NO_DEC1_FORK) == 0;
    // the lazy clock's slice of the previous step: in this call's catch-up launch (default), on the side stream, or at the end of its own step
    const bool defer_slice = fork_dec1 && (pp->flags & LTG_PIPE_NO_SLICE_FORK) == 0;
    // fork / join of the weight update: device words (ltg_pipe.sync) or event pairs
    const bool gates = fork_dec1 && pp->sync && (pp->flags & LTG_PIPE_EVENTS) == 0;
    // with device words the slice runs on the side stream between this call's catch-up and the next one's (two more words), beside
    // enc-1 / dec-0 instead of inside the catch-up launch on the critical stream
    const bool side_slice = gates && defer_slice && (pp->flags & LTG_PIPE_SLICE_IN_TOUCH) == 0;
    const bool merge_slice = defer_slice && !side_slice;   // (events, or LTG_PIPE_SLICE_IN_TOUCH: the slice rides in the catch-up launch)
    const bool ahead = side_slice && q0_ahead_capable(cfg, gen, bt, pp);   // catch-up of the NEXT batch's rows during this call (include/ltg.h)
    Workspace w = carve(cfg, B, nf, (char*)ws);
    if (o->y_pre) w.y = const_cast<float*>(o->y_pre);   // y_generated from ltg_fake_tower_batched
    const Probe pr{o->probe, st};
    const AdamC ad = make_adam(cfg, o->adam_t);
    float* rowpart = pp->rowpart_all + (size_t)rank * B * RP;   // this rank's block of the all-gather buffer: the exchange is in place
    // Device words of the pipe (ltg_pipe.sync; `seq` = this call's ordinal):
    //   0  the dh2 product of call seq is complete (set by the slab sum when it starts; polled by one wave in front of the weight update)
    //   1  the weight update of call seq has READ h2 (its workgroups count themselves in word 8 behind their prologue; ragged slabs: set
    //      when the update ends) -- polled by the last thread of the next call's enc-1, in front of dec-0, which overwrites h2
    //   7  the weight update of call seq has ENDED (one wave behind it) -- polled by the last thread of the next call's dec-0, in front of
    //      the streaming forward, which reads the shadow rows and the bias the update writes
    //   5  enc-0 of call seq has started = the catch-up of the batch's rows is complete (polled by one wave in front of the clock slice)
    //   6  the clock slice of call seq has ended (set by the next kernel of the side stream when it starts) -- polled by the last thread
    //      of call seq's OWN last kernel (the tail), in front of the next call's catch-up
    //   2  waits that gave up = the pipe's POISON: every kernel of the step that writes h2 or the model returns at once when it is set
    // Every poll is made by ONE thread, either a one-wave kernel of the side stream or the last thread of the kernel in front of the
    // one that needs the gate: no workgroup of a large launch ever holds a CU while it waits for a producer that still needs one.
    //   9  dh1 of call seq is complete (stored by the sparse gradient kernel when it starts; one wave in front of the Adam tail on the
    //      tail stream polls for it);  10  the tail of call seq has ended (one wave behind it) -- polled by the last thread of the next
    //      call's enc-0, in front of enc-1, which reads what the tail updates
    const unsigned* poison = (fork_dec1 && pp->sync && (pp->flags & LTG_PIPE_EVENTS) == 0) ? pp->sync + 2 : nullptr;
    // the Adam tail (W_p0, W_q1, the biases: it needs dh1's outputs only) on its OWN stream beside the sparse gradient kernel, the next
    // call's catch-up and enc-0; needs <= G0_LIGHT batch rows per partial bias row (its bias job then re-sums them in the same order)
    hipStream_t stl = (hipStream_t)pp->tail_stream;
    // OPT-IN (LTG_PIPE_TAIL_OWN).  It was the default without a communicator while the catch-up launch led the critical stream (20 000 items:
    // 146.0-147.1 against 150.9-151.9 us per step).  With the catch-up ahead and two shadow buffers the tail's own chain -- waiter, tail, bias
    // kernel, word 10 -- is what the next enc-0 then waits for: 20 000 items 142.0-143.6 on its own stream against 139.1-139.5 inline (equal,
    // 138.7-140.4 / 138.1-140.6, once the bias kernel was spread over 8 x as many threads), 200 000 items 802-825 against 745-767; with the
    // three RCCL calls in the stream (per-rank proxy) 154.4-157.0 against 154.6-155.1.
    const bool tail_own = gates && side_slice && stl && pp->ev_tail && (pp->flags & LTG_PIPE_TAIL_OWN) != 0 && (pp->flags & LTG_PIPE_WIDE_GRAD) == 0 &&
                          (B + ENC0_BIAS_PARTS - 1) / ENC0_BIAS_PARTS <= G0_LIGHT;
#define LTG_HIP(x) do { if ((x) != hipSuccess) return LTG_ELAUNCH; } while (0)
#define LTG_COMM(x) do { if ((x) != 0) return LTG_ELAUNCH; } while (0)
    // ---- the clock slice forked by the PREVIOUS call is done (it must not meet the catch-up below on a row); the rows this batch reads
    const int qP = gen->q0_period;
    if (merge_slice && gen->q0_ord > 0 && gen->q0_ord % qP < I) {
        const int start = gen->q0_ord % qP, ns = (I - start + qP - 1) / qP;
        hipLaunchKernelGGL(k_q0_touch_slice, dim3(bt->n_unique + ns), dim3(Q0_NT), 0, st, I, H, bt->n_unique, bt->uptr, bt->csr_pos, bt->indices, bt->uitem,
                           gen->q0_ord, start, qP, *gen, make_adam(cfg, 1));
    } else if (side_slice && ahead && pp->caught_up) {
        // the previous call brought this batch's rows up to q0_ord on the side stream (k_q0_touch_ahead; its end is behind word 6, which
        // that call's last kernel on this stream waited for) and marked them: no catch-up launch
    } else if (side_slice) {
        // the slice step t - 1 owes (rows i = ord (mod period) up to ord) on the SIDE stream, between this call's catch-up and the next call's:
        // word 5 is opened by enc-0 when it starts (the catch-up in front of it is complete: the rows of this batch are at ord, the slice
        // skips them whatever happens to them later), a one-wave kernel in front of the sweep polls for it; word 6 is opened by the side
        // stream's next kernel (the waiter in front of the weight update) when it starts, and the NEXT call's catch-up polls for it
        // (issued in the order the device needs them: the critical stream's kernels first)
        // (the previous call's last kernel waited for word 6 before it ended: ltg_gate_wait_tail in fk_g_tail)
        q0_touch(cfg, gen, bt, st, poison, ahead ? pp->q0_mark : nullptr, pp->seq);
    } else
        q0_touch(cfg, gen, bt, st, poison);
    // ---- forward: enc-0 over the local slab -> exchange 1 -> enc-1 (bias + tanh in its loader), dec-0, local logits + statistics
    // (Round 4, measured and removed: without a communicator, bias + tanh in enc-0 and the plain enc-1 behind it, h1 alternating between
    // two buffers: 143.5-147.2 against 145.2-148.1 us per step at 20 000 items, equal at 200 000 -- within the noise, one mode fewer.)
    {
        ltg_gen_acts a1 = *acts;
        a1.h1 = pp->h1pre;
        fwd_stage_enc(cfg, gen, bt, &o->fwd, &a1, 1, st, nullptr, true, side_slice ? LtgGate{pp->sync + 5, pp->seq, nullptr, 0} : LTG_NO_GATE,
                      tail_own ? LtgGate{pp->sync + 10, pp->seq - 1u, pp->sync + 2, 0} : LTG_NO_GATE);
    }
    // (Round 4, measured and removed: the clock's kernels on the pipe's THIRD stream instead of between two weight updates on the side
    // stream -- 142.8-143.4 against 141.4-142.4 us per step at 20 000 items, 155.3-156.7 against 155.1-157.2 at 25 024: no difference.  What
    // bounds the step at these sizes is the cycle update -> streaming forward -> dlogits -> dh2 product -> update, not either stream's load.)
    // (... nor did starting them late, behind word 0 on the third stream -- beside the weight update and the backward chain instead of beside
    // the streaming forward, the row statistics and dlogits: 142.2-142.7 against 139.2-139.7 us at 20 000 items, 159.2-159.4 against
    // 157.7-158.8 at 25 024, 831-833 against 823 at 200 000.)
    // (... and again with two shadow buffers in, when the side stream runs back to back -- update, slice, catch-up ahead, three one-wave
    // kernels = the step's length: the clock's kernels on the third stream then start earlier, beside the previous update's last third:
    // 143.1-147.3 against 137.8-138.9 us at 20 000 items, 153.5-154.3 against 152.2-153.5 at 25 024, 746-806 against 762-817 at 200 000.)
    if (side_slice) {
        const int start = gen->q0_ord % qP;
        hipStream_t sc = sd;
        hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, sc, LtgGate{pp->sync + 5, pp->seq, pp->sync + 2, 0}, LTG_NO_GATE);
        if (gen->q0_ord > 0 && start < I)
            hipLaunchKernelGGL(k_q0_sweep, dim3((I - start + qP - 1) / qP), dim3(Q0_NT), 0, sc, I, H, start, qP, gen->q0_ord, *gen, make_adam(cfg, 1), poison);
        if (ahead && pp->next_uitem && pp->next_nu > 0)
            hipLaunchKernelGGL(k_q0_touch_ahead, dim3(pp->next_nu), dim3(Q0_NT), 0, sc, H, pp->next_nu, pp->next_uitem, gen->q0_ord + 1, *gen, ad, pp->q0_mark,
                               pp->seq, poison);
    }
    if (comm) LTG_PROBED(pr, LTG_K_EXCH_H1, LTG_COMM(comm->all_reduce(pp->h1pre, pp->h1pre, (size_t)B * H, LTG_NCCL_FLOAT32, LTG_NCCL_SUM, comm->comm, stream)));
    LTG_PROBED(pr, LTG_K_ENC1, hipLaunchKernelGGL(fk_enc1<true>, grid2(Z, B, 16, 16), dim3(ENC1_NT), 0, st, B, H, Z, pp->h1pre, gen->p[1], gen->p[5], o->fwd.eps,
                                                  o->fwd.is_training, cfg->seed, o->fwd.rng_step, acts->mulv, acts->z, gen->p[4], acts->h1,
                                                  gates ? LtgGate{pp->sync + 1, pp->seq - 1u, pp->sync + 2, 0} : LTG_NO_GATE));
    // (dec-0 overwrites h2, which the previous step's weight update reads in its prologue: enc-1's last thread polled for word 1 -- or,
    // with events, the stream waits for the whole update; the streaming forward behind dec-0 needs the update's END: word 7)
    if (fork_dec1 && !gates) LTG_HIP(hipStreamWaitEvent(st, ev_dec1, 0));
    // (Round 4, measured and removed: slabs of 65 536 items or more with the weight update AND the streaming forward as two launches each
    // over the halves of the slab, the forward's first half beside the update's second -- bit-identical through carried per-lane softmax
    // statistics, but slower: 777-813 against 753-792 us per step at 200 000 items, same box.  Beside the update the forward's half takes
    // 290 us instead of 44 and the update 30 us longer: the step is HBM-bound, overlapping two bandwidth-bound kernels moves no byte
    // less.  profiles/r4_ab_c4_two_launch_split.txt, r4_c4_timeline_two_launch_split.txt.)
    LTG_PROBED(pr, LTG_K_DEC0, hipLaunchKernelGGL(fk_dec0, grid2(H, B, 16, 16), dim3(NT), 0, st, B, H, Z, acts->z, acts->mulv, gen->p[2], gen->p[6], acts->kl_rows,
                                                  acts->h2,
#ifdef LTG_X_NO_W7   // MEASUREMENT BUILD ONLY (results wrong): the streaming forward does not wait for the end of the previous weight update -- the
                     // upper bound of what ANY earlier hand-over of the shadow (tile by tile, word 7 sooner) could gain
                                                  LTG_NO_GATE,
#else
                                                  gates ? LtgGate{pp->sync + 7, pp->seq - 1u, pp->sync + 2, 0} : LTG_NO_GATE,
#endif
                                                  poison));
    // (Round 5, measured and removed: with ONE rank no exchange sits between the row statistics and dlogits, so k_row_stats_merge was folded
    // into k_dlogits_combine -- every (segment, row) workgroup re-folding the 256 (max, sum exp) pairs and the sparse terms of its row, the
    // step's scalars from the last of B tickets; bit-identical.  The fused kernel took 16.9 us against 10.0 + 5.6 for the two launches:
    // 143 against 139.5 us per step at 20 000 items, 102 against 101 at 25 024 -- the redundancy costs what the launch saved.)
    {
        int G = 0;
        LTG_PROBED(pr, LTG_K_DEC1_FWD, G = launch_dec1_fwd_stream(cfg, gen, B, acts, w.segpart, st));
        g_row_partial(cfg, bt, nf > 0 ? fake : nullptr, acts, rowpart, st, w.segpart, nullptr, G);
    }
    if (comm) LTG_PROBED(pr, LTG_K_EXCH_ROWPART, LTG_COMM(comm->all_gather(rowpart, pp->rowpart_all, (size_t)B * RP, LTG_NCCL_FLOAT32, comm->comm, stream)));
    // ---- backward: (fake tower when it was not evaluated ahead,) losses + dlogits
    if (nf > 0 && !o->y_pre && !(o->fake_done & 1)) {
        PairView pv{0, nf, nullptr, nullptr, fake->pop, fake->niche};
        DropView dA{nullptr, o->drop_fake[0], 0, 0}, dB{nullptr, o->drop_fake[1], 0, 0}, dC{nullptr, o->drop_fake[2], 0, 0};
        disc_forward(cfg, disc, pv, dA, dB, dC, o->d_keep_prob, o->d_rng_step, w, false, o->probe, st);
    }
    hipLaunchKernelGGL(k_dlogits_combine<true>, dim3((I + DL_SEG - 1) / DL_SEG, B), dim3(NT), 0, st, B, I, R, bt->indptr, bt->indices, bt->values, acts->logits,
                       pp->rowpart_all, acts->kl_rows, nf > 0 ? w.y : (const float*)nullptr, o->cnt, o->anneal, o->gan_lambda, nf, fake->row, fake->niche, fake->pop,
                       w.dlog, acts->lse, w.scal, loss_out, cfg->item_lo);
    {
        const int kchunk = dh2_stream_chunk(I), nsplit = (I + kchunk - 1) / kchunk;
        // Two shadow buffers (ltg_pipe.shadow_out): the update of this call writes the OTHER buffer, so the dh2 product -- the last reader of
        // this call's shadow -- leaves the cycle update -> streaming forward -> dlogits -> [dh2 product] -> update that bounds the step at
        // 20 000 - 25 000 items: the update's gate (word 0) opens when the product STARTS (dlogits is complete), not when it has ended.
        const bool pingpong = gates && shadow_pingpong(gen, pp);
        LTG_PROBED(pr, LTG_K_DH2, hipLaunchKernelGGL((k_dh2_stream<true, DH2_NH>), dim3(nsplit, DH2_NH), dim3(ST_NT), (size_t)2 * ST_BN * ST_LDW * 2, st, B, I, H, kchunk,
                                                     w.dlog, gen->wp1t_bf16, w.part, pingpong ? LtgGate{pp->sync, pp->seq, nullptr, 0} : LTG_NO_GATE));
        // ---- ONE fork, behind the dh2 product (the last reader of this step's W_p1t shadow), onto the side stream:
        //   (1) the clock slice of the PREVIOUS step -- rows i = q0_ord (mod period) up to q0_ord; every row of this batch is at q0_ord
        //       already (q0_touch), so the slice skips them whatever the rest of this step does to them; joined at the start of the next
        //       call, before that batch's catch-up;
        //   (2) the decoder weight update (needs dlogits and h2 only); joined before the NEXT step's dec-0
        ltg_g_opts od = *o;
        od.fake_done = 0;
        od.dec1_done = 0;
        const int dw_groups = (pp->flags >> 8) & 0x1FF;   // measurement: persistent workgroups of the weight update (0 = the library's choice)
        hipStream_t sdw = st;
        const int n_da2 = B * H;
        if (gates)   // the slab sum first: it is the next kernel of the critical stream, the side stream's launches take the host ~30 us
            hipLaunchKernelGGL(k_da2, dim3((n_da2 + NT - 1) / NT < 2048 ? (n_da2 + NT - 1) / NT : 2048), dim3(NT), 0, st, n_da2, nsplit, w.part, (const float*)nullptr,
                               pp->dh2, pingpong ? LTG_NO_GATE : LtgGate{pp->sync, pp->seq, nullptr, 0});
        if (gates) {   // the side stream's work starts behind a one-wave kernel that polls the word the slab sum (below) sets when it starts
            hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, sd, LtgGate{pp->sync, pp->seq, pp->sync + 2, 0},
                               side_slice ? LtgGate{pp->sync + 6, pp->seq, nullptr, 0} : LTG_NO_GATE);
            sdw = sd;
        } else if (fork_dec1) {
            LTG_HIP(hipEventRecord(ev_fork, st));
            LTG_HIP(hipStreamWaitEvent(sd, ev_fork, 0));
            sdw = sd;
        }
        // (a ragged slab's last I % 32 rows go through the generic tile kernel behind the streaming one, and that reads h2 throughout:
        // word 1 then opens with word 7)
        const bool h2_early = gates && (I % 32) == 0;
        // (Round 4, measured and removed: word 7 stored by the update's LAST workgroup -- every thread releases its stores at agent scope,
        // the workgroups count themselves -- instead of by a kernel behind it: 142 -> 184 us per step at 20 000 items, 155 -> 195 at
        // 25 024, 749 -> 856 at 200 000: 1 256 waves each writing the dirty lines of an L2 back cost far more than the ~4 us the word opens earlier.)
        const LtgH2Done hd_last = h2_early ? LtgH2Done{pp->sync + 8, pp->sync + 1, pp->seq, poison} : LtgH2Done{nullptr, nullptr, 0u, poison};
        ltg_gen_state gen_dw = *gen;
        if (pingpong) gen_dw.wp1t_bf16 = pp->shadow_out;   // (the caller exchanges the two pointers after the call)
        const int rc = g_stage_bwd_rest(cfg, &gen_dw, bt, &od, acts, nullptr, w, sdw, true, true, dw_groups, hd_last);
        if (rc != LTG_OK) return rc;
        // (Round 4, measured and removed: word 7 stored by the NEXT call's first waiter on the side stream when it starts, instead of by one
        // wave behind the update: 149.8 against 146.8 us per step at 20 000 items, 165.7 against 161.5 at 25 024 -- slower; and a host that is
        // not launches ahead of the device, a two-rank rig with host-side exchanges, stalls dec-0 on it.)
        if (gates)
            hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, sd, LtgGate{pp->sync + 7, pp->seq, nullptr, 0},
                               h2_early ? LTG_NO_GATE : LtgGate{pp->sync + 1, pp->seq, nullptr, 0});
        else if (fork_dec1) LTG_HIP(hipEventRecord(ev_dec1, sd));
        if (!gates)
            hipLaunchKernelGGL(k_da2, dim3((n_da2 + NT - 1) / NT < 2048 ? (n_da2 + NT - 1) / NT : 2048), dim3(NT), 0, st, n_da2, nsplit, w.part, (const float*)nullptr,
                               pp->dh2);
    }
    if (comm) LTG_PROBED(pr, LTG_K_EXCH_DH2, LTG_COMM(comm->all_reduce(pp->dh2, pp->dh2, (size_t)B * H, LTG_NCCL_FLOAT32, LTG_NCCL_SUM, comm->comm, stream)));
    // ---- the replicated rest: dz (tanh derivative in its loader) -> dh1 -> sparse W_q0 gradient + its Adam step -> the other updates
    LTG_PROBED(pr, LTG_K_DZ, hipLaunchKernelGGL(fk_dz_dh2, grid2(Z, B, 16, 16), dim3(NT), 0, st, B, Z, H, pp->dh2, acts->h2, gen->p[2], acts->mulv, o->fwd.eps,
                                                o->fwd.is_training, o->anneal, cfg->seed, o->fwd.rng_step, w.dmlv, w.da2));
    LTG_PROBED(pr, LTG_K_DH1, hipLaunchKernelGGL(fk_dh1, grid2(H, B, 16, 16), dim3(NT), 0, st, B, H, 2 * Z, w.dmlv, gen->p[1], acts->h1, w.da1));
    const LtgGate slice_done = side_slice ? LtgGate{pp->sync + 6, pp->seq, pp->sync + 2, 0} : LTG_NO_GATE;
    if (tail_own) {
        g_enc0_grad(cfg, bt, o, acts, w, st, gen, &ad, true, poison, LtgGate{pp->sync + 9, pp->seq, nullptr, 0}, slice_done,
                    gen->q0_lr_hist + ((gen->q0_ord + 1) & (LTG_Q0_HIST - 1)));
        hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, stl, LtgGate{pp->sync + 9, pp->seq, pp->sync + 2, 0}, LTG_NO_GATE);
        g_jobs(-1, cfg, gen, bt, o, acts, w, ad, nullptr, false, nullptr, stl, true, true, poison, LTG_NO_GATE, true);
        hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, stl, LtgGate{pp->sync + 10, pp->seq, nullptr, 0}, LTG_NO_GATE);
    } else {
        g_enc0_grad(cfg, bt, o, acts, w, st, gen, &ad, (pp->flags & LTG_PIPE_WIDE_GRAD) == 0, poison);
        g_jobs(-1, cfg, gen, bt, o, acts, w, ad, nullptr, false, nullptr, st, true, true, poison, slice_done);
    }
    if (!defer_slice) {   // the slice of THIS step at its end, in program order (the cut-point schedule)
        const int ord = gen->q0_ord + 1, start = ord % qP;
        if (start < I) hipLaunchKernelGGL(k_q0_sweep, dim3((I - start + qP - 1) / qP), dim3(Q0_NT), 0, st, I, H, start, qP, ord, *gen, make_adam(cfg, 1));
    }
#undef LTG_HIP
#undef LTG_COMM
    return check_launch();
}

int ltg_refresh_d_shadow(const ltg_config* cfg, const ltg_disc_state* d, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !d || !d->emb || !d->emb_fp8 || !d->w1t_fp8 || !d->w2t_fp8 || !d->w3t_fp8) return LTG_EINVAL;
    hipLaunchKernelGGL(k_d_shadow, dim3(2048), dim3(NT), 0, (hipStream_t)stream, cfg->d_feat, cfg->d_h0, cfg->d_h1, cfg->d_h2, cfg->d_h3, d->emb, d->p[0],
                       d->p[2], d->p[4], const_cast<uint8_t*>(d->emb_fp8), d->w1t_fp8, d->w2t_fp8, d->w3t_fp8, d->w3_fp8);
    return check_launch();
}

int ltg_g_flush(const ltg_config* cfg, const ltg_gen_state* gen, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !gen || !q0_lazy(cfg, gen)) return LTG_EINVAL;
    const int I = cfg->n_items;
    hipLaunchKernelGGL(k_q0_sweep, dim3(I < 65536 ? I : 65536), dim3(Q0_NT), 0, (hipStream_t)stream, I, cfg->h_enc, 0, 1, gen->q0_ord, *gen, make_adam(cfg, 1));
    return check_launch();
}

int ltg_refresh_shadow(const ltg_config* cfg, const ltg_gen_state* gen, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !gen || !gen->wp1t_bf16 || cfg->h_enc > ST_KP) return LTG_EINVAL;
    const size_t total = (size_t)cfg->n_items * ST_KP;
    size_t gx = (total + NT - 1) / NT;
    if (gx > 65536) gx = 65536;
    hipLaunchKernelGGL(k_refresh_shadow, dim3((unsigned)gx), dim3(NT), 0, (hipStream_t)stream, cfg->n_items, cfg->h_enc, gen->p[3], gen->wp1t_bf16);
    return check_launch();
}

int ltg_gather_cand_logits(const ltg_config* cfg, const ltg_sample_inputs* in, const float* logits, float* out, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !in || !logits || !out || in->n_rows < 0) return LTG_EINVAL;
    if (in->n_rows == 0) return LTG_OK;
    hipLaunchKernelGGL(k_gather_cand, dim3(in->n_rows), dim3(NT), 0, (hipStream_t)stream, cfg->n_items, cfg->item_lo, in->cand_ptr,
                       in->cand_idx, logits, out);
    return check_launch();
}

int ltg_rank_metrics(const ltg_config* cfg, const float* logits, const ltg_batch* tr, const ltg_batch* te, int32_t k_ndcg,
                     int32_t k_r1, int32_t k_r2, float* out, ltg_stream stream) {
    clear_errors();
    if (!cfg_ok(cfg) || !logits || !tr || !te || !out || tr->n_rows != te->n_rows || tr->n_rows < 0) return LTG_EINVAL;
    if (tr->n_rows == 0) return LTG_OK;
    const size_t lds = (size_t)((cfg->n_items + 31) / 32) * sizeof(unsigned);
    if (lds > 64 * 1024) return LTG_EINVAL;
    hipLaunchKernelGGL(k_rank_metrics, dim3(tr->n_rows), dim3(NT), lds, (hipStream_t)stream, cfg->n_items, cfg->item_lo, logits,
                       tr->indptr, tr->indices, te->indptr, te->indices, (const float*)nullptr, (int32_t*)nullptr, k_ndcg, k_r1, k_r2, out);
    return check_launch();
}

int ltg_fp8_roundtrip(const float* in, float* out, int32_t n, ltg_stream stream) {
    clear_errors();
    if (!in || !out || n < 0) return LTG_EINVAL;
    if (n == 0) return LTG_OK;
    hipLaunchKernelGGL(k_fp8_roundtrip, dim3((n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024), dim3(NT), 0, (hipStream_t)stream, n, in, out);
    return check_launch();
}

int ltg_debug_gemm(int32_t mode, int32_t M, int32_t N, int32_t K, const float* A, const float* B, float* C, ltg_stream stream) {
    clear_errors();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || mode < 0 || mode > 2) return LTG_EINVAL;
    const dim3 g((N + 31) / 32, (M + 31) / 32);
    if (mode == 0) hipLaunchKernelGGL(k_debug_gemm<0>, g, dim3(NT), 0, (hipStream_t)stream, M, N, K, A, B, C);
    else if (mode == 1) hipLaunchKernelGGL(k_debug_gemm<1>, g, dim3(NT), 0, (hipStream_t)stream, M, N, K, A, B, C);
    else hipLaunchKernelGGL(k_debug_gemm<2>, g, dim3(NT), 0, (hipStream_t)stream, M, N, K, A, B, C);
    return check_launch();
}

int ltg_rank_scores(const ltg_config* cfg, const float* logits, const ltg_batch* tr, const ltg_batch* te, float* score_out,
                    ltg_stream stream) {
    clear_errors();
    if (!cfg || !logits || !tr || !te || !score_out || tr->n_rows != te->n_rows) return LTG_EINVAL;
    if (tr->n_rows == 0) return LTG_OK;
    hipLaunchKernelGGL(k_rank_scores, dim3(tr->n_rows), dim3(NT), 0, (hipStream_t)stream, cfg->n_items, cfg->item_lo, tr->n_rows, logits,
                       tr->indptr, tr->indices, te->indptr, te->indices, score_out);
    return check_launch();
}

int ltg_rank_counts(const ltg_config* cfg, const float* logits, const ltg_batch* tr, const ltg_batch* te, const float* score,
                    int32_t* count_out, ltg_stream stream) {
    clear_errors();
    if (!cfg || !logits || !tr || !te || !score || !count_out || tr->n_rows != te->n_rows) return LTG_EINVAL;
    if (tr->n_rows == 0) return LTG_OK;
    const size_t lds = (size_t)((cfg->n_items + 31) / 32) * sizeof(unsigned);
    if (lds > 64 * 1024) return LTG_EINVAL;
    hipLaunchKernelGGL(k_rank_metrics, dim3(tr->n_rows), dim3(NT), lds, (hipStream_t)stream, cfg->n_items, cfg->item_lo, logits,
                       tr->indptr, tr->indices, te->indptr, te->indices, score, count_out, 0, 0, 0, (float*)nullptr);
    return check_launch();
}

int ltg_rank_finish(const ltg_batch* te, const int32_t* counts, int32_t k_ndcg, int32_t k_r1, int32_t k_r2, float* out,
                    ltg_stream stream) {
    clear_errors();
    if (!te || !counts || !out) return LTG_EINVAL;
    if (te->n_rows == 0) return LTG_OK;
    hipLaunchKernelGGL(k_rank_finish, dim3((te->n_rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, te->n_rows, te->indptr, counts,
                       k_ndcg, k_r1, k_r2, out);
    return check_launch();
}

#ifdef LTG_STAMP
// MEASUREMENT BUILD ONLY: copies the phase stamps of the stamped kernel's last launch (ltg_rgemm.h) to the host
int ltg_debug_stamps(void* dst, int n_words) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ltg_stamp_buf), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
}  // extern "C"
