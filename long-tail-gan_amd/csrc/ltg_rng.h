// Counter-based RNG shared bit-for-bit with oracle/ltg_oracle.py (rng_u64 / rng_uniform / rng_normal).
// Every random tensor of the path (tf.nn.dropout masks MultiVAE.py:149 / discriminator.py:25,30,44;
// tf.random_normal MultiVAE.py:178; np.random.choice sample.py:54, train.py:236) is a pure function
// of (seed, stream, step, element index), so a step is reproducible and shardable.
#pragma once
#include <stdint.h>

#define LTG_STREAM_VAE_DROPOUT 1
#define LTG_STREAM_VAE_EPS 2
#define LTG_STREAM_D_DROP_A 3
#define LTG_STREAM_D_DROP_B 4
#define LTG_STREAM_D_DROP_C 5
#define LTG_STREAM_GUMBEL 6
#define LTG_STREAM_POP_PICK 7

__device__ __forceinline__ uint64_t ltg_rng_u64(uint64_t seed, uint32_t stream, uint64_t step, uint64_t idx) {
    uint64_t z = (seed ^ ((uint64_t)stream * 0xD6E8FEB86659FD93ull)) + step * 0x94D049BB133111EBull;
    z += (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// uniform in [0,1), 24 bits
__device__ __forceinline__ float ltg_rng_uniform(uint64_t seed, uint32_t stream, uint64_t step, uint64_t idx) {
    return (float)(ltg_rng_u64(seed, stream, step, idx) >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float ltg_rng_normal(uint64_t seed, uint32_t stream, uint64_t step, uint64_t idx) {
    const uint64_t z = ltg_rng_u64(seed, stream, step, idx);
    const float u1 = ((float)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)((z >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// keep decision of tf.nn.dropout (P(keep) = keep_prob)
__device__ __forceinline__ bool ltg_rng_keep(uint64_t seed, uint32_t stream, uint64_t step, uint64_t idx, float keep) {
    return ltg_rng_uniform(seed, stream, step, idx) < keep;
}
