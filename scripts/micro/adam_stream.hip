// How many CUs does an Adam sweep (24 B per parameter: theta / m / v read and written once) need to reach the HBM rate?
// Persistent workgroups, U float4 per lane and tensor in flight (two register sets: the next block is requested before the
// current one is consumed), non-temporal accesses -- the memory side of k_dec1_bwd_adam_stream without its product.
//   hipcc -O3 --offload-arch=gfx950 -o build_ab/adam_stream scripts/micro/adam_stream.hip ; ./build_ab/adam_stream [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float __attribute__((ext_vector_type(4))) f4;

__device__ __forceinline__ float move(float p, float lm, float v) {
    return __builtin_fmaf(-lm, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + 1e-8f), p);
}
__device__ __forceinline__ void adam4(f4& p, f4& m, f4& v, float g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = __builtin_fmaf(0.9f, m[i], 0.1f * g);
        v[i] = __builtin_fmaf(0.999f, v[i], (0.001f * g) * g);
        p[i] = move(p[i], 1e-4f * m[i], v[i]);
    }
}

// block b = float4 [b * T * U, (b + 1) * T * U): lane's element j of it = b * T * U + j * T + tid (1-KiB wave accesses)
template <int U, bool NT>
__global__ void k_sweep(size_t n4, f4* __restrict__ W, f4* __restrict__ M, f4* __restrict__ V, float g) {
    const size_t T = blockDim.x, per = T * U, nb = n4 / per;
    f4 p[2][U], m[2][U], v[2][U];
    auto ld = [&](int s, size_t b) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t e = b * per + j * T + threadIdx.x;
            if (NT) { p[s][j] = __builtin_nontemporal_load(W + e); m[s][j] = __builtin_nontemporal_load(M + e); v[s][j] = __builtin_nontemporal_load(V + e); }
            else { p[s][j] = W[e]; m[s][j] = M[e]; v[s][j] = V[e]; }
        }
    };
    auto st = [&](int s, size_t b) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t e = b * per + j * T + threadIdx.x;
            adam4(p[s][j], m[s][j], v[s][j], g);
            if (NT) { __builtin_nontemporal_store(p[s][j], W + e); __builtin_nontemporal_store(m[s][j], M + e); __builtin_nontemporal_store(v[s][j], V + e); }
            else { W[e] = p[s][j]; M[e] = m[s][j]; V[e] = v[s][j]; }
        }
    };
    size_t b = blockIdx.x;
    if (b >= nb) return;
    ld(0, b);
    for (;;) {
        const size_t b1 = b + gridDim.x;
        if (b1 < nb) ld(1, b1);
        __builtin_amdgcn_sched_barrier(0);
        st(0, b);
        if (b1 >= nb) break;
        const size_t b2 = b1 + gridDim.x;
        if (b2 < nb) ld(0, b2);
        __builtin_amdgcn_sched_barrier(0);
        st(1, b1);
        if (b2 >= nb) break;
        b = b2;
    }
}

template <int U, bool NT>
static double run(int G, int T, size_t n4, f4* W, f4* M, f4* V, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_sweep<U, NT>), dim3(G), dim3(T), 0, s, n4, W, M, V, 1e-3f);
    double best = 1e30, sum = 0;
    const int reps = 8;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL((k_sweep<U, NT>), dim3(G), dim3(T), 0, s, n4, W, M, V, 1e-3f);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    (void)sum;
    return best;
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? atol(argv[1]) : 25024, H = 600;
    const size_t n4 = rows * H / 4;
    f4 *W, *M, *V;
    hipMalloc(&W, n4 * 16);
    hipMalloc(&M, n4 * 16);
    hipMalloc(&V, n4 * 16);
    hipMemset(W, 0, n4 * 16);
    hipMemset(M, 0, n4 * 16);
    hipMemset(V, 0x3c, n4 * 16);
    hipStream_t s;
    hipStreamCreate(&s);
    const double bytes = 24.0 * rows * H;
    printf("rows %zu: %.1f MB per sweep (read + write)\n", rows, bytes / 1e6);
    const int Gs[] = {64, 96, 128, 160, 196, 256, 512};
    const int Ts[] = {256, 512, 1024};
    printf("%-22s", "threads x U (nt)");
    for (int G : Gs) printf(" %8d", G);
    printf("   <- workgroups; cells: TB/s (GB/s per workgroup)\n");
    for (int T : Ts) {
        for (int U = 1; U <= 4; U *= 2) {
            for (int nt = 1; nt >= 0; --nt) {
                if (!nt && !(T == 512 && U == 2)) continue;
                printf("%4d x %d %-12s", T, U, nt ? "nt" : "plain");
                for (int G : Gs) {
                    double ms;
                    if (U == 1) ms = nt ? run<1, true>(G, T, n4, W, M, V, s) : run<1, false>(G, T, n4, W, M, V, s);
                    else if (U == 2) ms = nt ? run<2, true>(G, T, n4, W, M, V, s) : run<2, false>(G, T, n4, W, M, V, s);
                    else ms = nt ? run<4, true>(G, T, n4, W, M, V, s) : run<4, false>(G, T, n4, W, M, V, s);
                    printf(" %5.2f(%2.0f)", bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / G);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
