// One-shot exchange over peer-mapped buffers: a SECOND transport behind ltg_comm, correctness only (round 6; SURVEY 8/e1: "messages <= 240 KB
// are latency-bound -> single-shot direct reduce-scatter + all-gather using all 7 links rather than a ring").  RCCL stays the default and
// the transport every number of this repository was measured with; this one exists so that the A/B is ready for the first box with more
// than one GPU.  Same entry-point signatures as ncclAllReduce / ncclAllGather (ltg_comm.all_reduce / .all_gather), so ltg_g_step_sharded
// issues it in-stream exactly like an RCCL collective.
//
// Every rank owns ONE staging buffer (ltg_oneshot_stage_bytes; device memory the caller allocates and shares with hipIpcGetMemHandle /
// hipIpcOpenMemHandle -- ltgan/_rccl.py: OneShotComm), laid out as
//     u32   flags [2][n_ranks]          flags[par][q]: rank q's message of the exchange with parity `par` has arrived (= its ordinal)
//     u32   done  [2], expired          block counter of this rank's pushing kernel; waits that gave up
//     float data  [2][n_ranks][max_floats]
// An exchange with ordinal s (parity s & 1) is ONE kernel per rank, one hop:
//   push   every element of my message into slot [par][my rank] of EVERY peer's buffer (system-scope write-through stores), release, and -- the
//          last of my blocks -- store s into flags[par][my rank] at every peer;
//   wait   one thread per block polls MY flags[par][q] for every peer q (bounded: limit_ms, then `expired` counts and the host raises);
//   sum    recv[i] = sum over q = 0 .. n_ranks - 1 IN RANK ORDER of (q == me ? mine : slot[par][q][i]) -- a fixed order, so every rank ends with
//          the same bits (all-gather: recv[q][i] = slot[par][q][i]).
// Why two parities suffice: rank r pushes exchange s + 2 into the slots of parity s only after it has finished exchange s + 1, which needed
// every peer's push of s + 1, which a peer issues -- stream order -- only behind its own kernel of exchange s, i.e. after it has read the slots.
// The grid is at most 64 blocks, all resident, and no block waits for another block of its own kernel: no deadlock by construction.
#pragma once

constexpr int OS_NT = 256, OS_MAX_BLOCKS = 64;
__host__ __device__ inline size_t os_flags_words(int R) { return (size_t)2 * R + 4; }
__host__ __device__ inline size_t os_data_off(int R) { return (os_flags_words(R) * 4 + 255) / 256 * 256; }

struct OsView {
    char* stage[LTG_ONESHOT_MAX_RANKS];
    int R, rank;
    unsigned seq, limit_ms;
    size_t max_floats;
};
__device__ __forceinline__ unsigned* os_flags(char* st) { return reinterpret_cast<unsigned*>(st); }
__device__ __forceinline__ float* os_slot(char* st, int R, size_t max_floats, int par, int q) {
    return reinterpret_cast<float*>(st + os_data_off(R)) + ((size_t)par * R + q) * max_floats;
}

// GATHER: recv = [R][count] (all-gather) instead of the rank-ordered sum
template <bool GATHER>
__global__ __launch_bounds__(OS_NT) void k_oneshot_exchange(OsView v, const float* send, float* recv, size_t count) {
    const int par = (int)(v.seq & 1u), R = v.R, me = v.rank;
    const size_t stride = (size_t)gridDim.x * OS_NT, i0 = (size_t)blockIdx.x * OS_NT + threadIdx.x;
    // push
    for (size_t i = i0; i < count; i += stride) {
        const float x = send[i];
        for (int p = 0; p < R; ++p)
            if (p != me) __hip_atomic_store(os_slot(v.stage[p], R, v.max_floats, par, me) + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: the stores above are in memory before the flag
    __syncthreads();
    unsigned* mine = os_flags(v.stage[me]);
    if (threadIdx.x == 0) {
        unsigned* done = mine + 2 * R + par;
        const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == gridDim.x) {                    // the last block of this rank's push
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = 0; p < R; ++p)
                if (p != me) __hip_atomic_store(os_flags(v.stage[p]) + par * R + me, v.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // wait for every peer's message of this exchange
        const unsigned long long ticks = (unsigned long long)(v.limit_ms > 0 ? v.limit_ms : 30000u) * 100000ull, t0 = wall_clock64();
        bool gave_up = false;
        for (int q = 0; q < R && !gave_up; ++q) {
            if (q == me) continue;
            while ((int)(__hip_atomic_load(mine + par * R + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - v.seq) < 0) {
                if (wall_clock64() - t0 > ticks) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (gave_up) atomicAdd(mine + 2 * R + 2, 1u);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // sum in rank order / gather
    for (size_t i = i0; i < count; i += stride) {
        const float x = send[i];
        if constexpr (GATHER) {
            for (int q = 0; q < R; ++q)
                recv[(size_t)q * count + i] =
                    q == me ? x : __hip_atomic_load(os_slot(v.stage[me], R, v.max_floats, par, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            float acc = 0.f;
            for (int q = 0; q < R; ++q)
                acc += q == me ? x : __hip_atomic_load(os_slot(v.stage[me], R, v.max_floats, par, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            recv[i] = acc;
        }
    }
}
