// Part of csrc/ltg_kernels.hip (one translation unit, one anonymous namespace; included there in this order): the streaming (HBM-bound) decoder kernels of large item slabs: forward (two forms), dh2 product, weight update + Adam + shadow refresh.
// Split out of the 4 400-line file in round 6 -- the code is unchanged.
#pragma once

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
