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


#include "ltg_gen_fwd.h"
#include "ltg_stream.h"
#include "ltg_disc.h"
#include "ltg_gstep.h"
#include "ltg_clock.h"
#include "ltg_sampler.h"

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
#include "ltg_oneshot.h"
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
    // (the four-term form is cheap enough to pay in fk_d_l1 too: D phase 45.5 against 45.9 ms, profiles/r6_ab_d_arith.txt)
    const int mode = c->d_arith & 3, set = ((c->d_arith >> 4) & 15) ? ((c->d_arith >> 4) & 15) : (mode == LTG_DARITH_BF16X4 ? 0xF : LTG_D_SPLIT_SET);
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
    if (!with_bwd && ft_wsp_capable(cfg) && (cfg->d_arith & 3) != LTG_DARITH_FP32 && (cfg->tuning & (1 << 20)) == 0) {
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
    if (d_fast(cfg) && !with_bwd && h0 >= 32 && h12 >= 32 && (h0 % 4) == 0 && (h12 % 4) == 0) {
        // forward only (the fake tower of the G steps, one batch or -- ltg_fake_tower_batched -- 10^5 pair rows): LDS-staged 64 x 64
        // tiles.  The choice depends on the layer sizes only, so the tower inside a step and the batched tower run the same
        // kernels and produce the same bits.  (d_arith = fp32, tuning bit 20, or sizes the one-kernel tower does not take.)
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
#ifndef LTG_D_FORK_MIN_ROWS
#define LTG_D_FORK_MIN_ROWS 1024
#endif
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
        // (Round 6: only from LTG_D_FORK_MIN_ROWS pair rows.  At the ~300 rows per step of the synthetic tables the two jobs are one or two slabs of work and
        // the fork's two gate kernels and its poll cost more than they hide -- and not steadily: D step 32.6-38.9 us with the fork against 34.1 +- 0.1
        // without at 20 000 items, 34.1-38.2 against 32.4 at 200 000 -- profiles/r6_ab_d_fork_small.txt.  Either way the same kernels write the same slab
        // entries: same bits.)
        const bool fork = o->aux_stream && o->sync && !grad_out && (cfg->tuning & 64) == 0 &&      // (tuning-knob bit 6: no fork)
                          n >= LTG_D_FORK_MIN_ROWS && d_adam_flat(cfg, disc, L, SP, w.slab, nullptr);
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

/* ---- one-shot exchange over peer-mapped staging buffers (csrc/ltg_oneshot.h) ---- */
size_t ltg_oneshot_stage_bytes(int32_t n_ranks, size_t max_floats) {
    if (n_ranks < 1 || n_ranks > LTG_ONESHOT_MAX_RANKS) return 0;
    return os_data_off(n_ranks) + (size_t)2 * n_ranks * max_floats * sizeof(float);
}
size_t ltg_oneshot_expired_offset(int32_t n_ranks) { return ((size_t)2 * n_ranks + 2) * sizeof(uint32_t); }
static int oneshot_exchange(bool gather, const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, ltg_stream stream) {
    ltg_oneshot* os = static_cast<ltg_oneshot*>(comm);
    if (!os || !sendbuf || !recvbuf || dtype != LTG_NCCL_FLOAT32 || (!gather && op != LTG_NCCL_SUM) || os->n_ranks < 1 || os->n_ranks > LTG_ONESHOT_MAX_RANKS ||
        os->rank < 0 || os->rank >= os->n_ranks || count > os->max_floats)
        return LTG_EINVAL;
    if (count == 0) return LTG_OK;
    OsView v;
    for (int q = 0; q < LTG_ONESHOT_MAX_RANKS; ++q) v.stage[q] = q < os->n_ranks ? static_cast<char*>(os->stage[q]) : nullptr;
    for (int q = 0; q < os->n_ranks; ++q)
        if (!v.stage[q]) return LTG_EINVAL;
    os->seq += 1u;
    v.R = os->n_ranks;
    v.rank = os->rank;
    v.seq = os->seq;
    v.limit_ms = os->limit_ms;
    v.max_floats = os->max_floats;
    const int nb = (int)((count + OS_NT - 1) / OS_NT);
    const dim3 g(nb < OS_MAX_BLOCKS ? nb : OS_MAX_BLOCKS);
    // (in place: the all-gather's own block is written from `send` -- which may BE that block -- element by element by the thread that read it)
    if (gather) hipLaunchKernelGGL(k_oneshot_exchange<true>, g, dim3(OS_NT), 0, (hipStream_t)stream, v, (const float*)sendbuf, (float*)recvbuf, count);
    else hipLaunchKernelGGL(k_oneshot_exchange<false>, g, dim3(OS_NT), 0, (hipStream_t)stream, v, (const float*)sendbuf, (float*)recvbuf, count);
    return hipGetLastError() == hipSuccess ? LTG_OK : LTG_ELAUNCH;
}
int ltg_oneshot_all_reduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, ltg_stream stream) {
    return oneshot_exchange(false, sendbuf, recvbuf, count, dtype, op, comm, stream);
}
int ltg_oneshot_all_gather(const void* sendbuf, void* recvbuf, size_t sendcount, int dtype, void* comm, ltg_stream stream) {
    return oneshot_exchange(true, sendbuf, recvbuf, sendcount, dtype, LTG_NCCL_SUM, comm, stream);
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

int ltg_debug_split(const float* in, float* out, int32_t n, ltg_stream stream) {
    clear_errors();
    if (!in || !out || n < 0 || (n % 4) != 0) return LTG_EINVAL;
    if (n == 0) return LTG_OK;
    const int n4 = n / 4;
    hipLaunchKernelGGL(k_debug_split, dim3((n4 + NT - 1) / NT < 1024 ? (n4 + NT - 1) / NT : 1024), dim3(NT), 0, (hipStream_t)stream, n4,
                       reinterpret_cast<const ltg_f32x4*>(in), out);
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
