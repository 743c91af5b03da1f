// What launching costs the HOST as a function of a kernel's shape (round 5): back-to-back eager launches of a kernel that does (almost) nothing,
// with the grids, LDS footprints, register counts and kernel-argument sizes of the path's latency kernels.  us per launch = HIP events around
// 2 000 launches on one stream.  Result (profiles/r5_micro_launch_rate.txt): 2.6-3.5 us per launch whatever the grid (13 .. 1 600 workgroups), LDS
// or registers -- the figure is the host's eager launch rate, not a device-side envelope -- and it grows with the ARGUMENT bytes (16 B: 2.7 us,
// 256 B: 3.3 us, 768 B: 3.4 us).  The training loop issues 9 / 6 launches per 68 / 54-us G / D step: the host stays ahead.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_envelope scripts/micro/launch_envelope.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NB> struct Blob { unsigned char b[NB]; };

template <int LDSF, int NB, int REGS>
__global__ __launch_bounds__(256) void k_empty(float* p, Blob<NB> blob, int never) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    float r[REGS];
    if (never) {      // keeps LDS, the blob and the registers alive without executing anything
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = p[i * 64 + threadIdx.x];
        __syncthreads();
        float s = blob.b[threadIdx.x % NB];
#pragma unroll
        for (int i = 0; i < REGS; ++i) s += r[i] * r[(i + 1) % REGS];
        lds[threadIdx.x % (LDSF > 0 ? LDSF : 1)] = s;
        __syncthreads();
        p[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) % (LDSF > 0 ? LDSF : 1)];
    }
}
// the same with ONE store per thread (dirty lines for the end-of-kernel write-back) and one load at entry (the previous launch's output)
template <int LDSF, int NB>
__global__ __launch_bounds__(256) void k_touch(float* p, float* q, Blob<NB> blob, int never) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x = p[i];
    if (never) { lds[threadIdx.x % (LDSF > 0 ? LDSF : 1)] = blob.b[0]; __syncthreads(); x += lds[0]; }
    q[i] = x + 1.f;
}

template <class F> double timeit(F launch, int reps = 2000) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 64; ++i) launch(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch(i);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
}
int main() {
    float *p, *q;
    (void)hipMalloc(&p, 64 << 20); (void)hipMalloc(&q, 64 << 20);
    (void)hipMemset(p, 0, 64 << 20); (void)hipMemset(q, 0, 64 << 20);
    printf("us per dependent launch, one stream (kernel body never executes unless noted)\n");
    printf("%-58s %8s %8s %8s %8s\n", "shape", "13 WGs", "100", "580", "1600");
#define ROW(NAME, K, ...)                                                                                                   \
    {                                                                                                                       \
        printf("%-58s", NAME);                                                                                              \
        for (int g : {13, 100, 580, 1600}) printf(" %8.2f", timeit([&](int) { hipLaunchKernelGGL(K, dim3(g), dim3(256), 0, 0, __VA_ARGS__); })); \
        printf("\n");                                                                                                       \
    }
    ROW("no LDS, 16-B arguments, few registers", (k_empty<0, 8, 4>), p, Blob<8>{}, 0)
    ROW("18 KB LDS", (k_empty<4608, 8, 4>), p, Blob<8>{}, 0)
    ROW("18 KB LDS, 256-B arguments", (k_empty<4608, 256, 4>), p, Blob<256>{}, 0)
    ROW("18 KB LDS, 768-B arguments", (k_empty<4608, 768, 4>), p, Blob<768>{}, 0)
    ROW("18 KB LDS, 768-B arguments, ~128 registers", (k_empty<4608, 768, 96>), p, Blob<768>{}, 0)
    ROW("no LDS, 16-B arguments, load + store per thread (ping-pong)", (k_touch<0, 8>), p, q, Blob<8>{}, 0)
    ROW("18 KB LDS, 768-B arguments, load + store per thread", (k_touch<4608, 768>), p, q, Blob<768>{}, 0)
    return 0;
}
