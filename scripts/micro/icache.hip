// Cost of COLD instruction fetch for short launches (round 5): the path's kernels are 2-26 KB of mostly straight-line code, run by one workgroup
// per CU for 5-15 us, and a step's kernels together exceed the 64-KB instruction cache two CUs share.  What does a KB of cold code cost?
//   straight<ID, KB>: KB kilobytes of straight-line v_fma (8 bytes each, a dependent chain -> 1 instruction per ~4-8 cycles per wave)
//   looped<ID>:       the same number of instructions from a 512-byte loop body
// 16 distinct instantiations of each are launched round-robin (16 x KB > 64 KB for KB >= 8: every launch finds the cache cold);
// the same instantiation launched back to back gives the warm figure.  Grid: 1 workgroup of 256 threads per CU (256 workgroups), or 32.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/icache scripts/micro/icache.hip ; run: /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA8(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" \
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))

template <int ID, int KB>
__global__ __launch_bounds__(256) void straight(float* p, float a, float b) {
    float x = p[threadIdx.x] + ID;
#pragma unroll
    for (int i = 0; i < KB * 16; ++i) FMA8(x);     // 16 x 8 instructions x 8 B = 1 KB
    p[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int ID>
__global__ __launch_bounds__(256) void looped(float* p, float a, float b, int kb) {
    float x = p[threadIdx.x] + ID;
    for (int i = 0; i < kb * 2; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) FMA8(x);       // 512-byte body
    }
    p[blockIdx.x * 256 + threadIdx.x] = x;
}

typedef void (*launch_t)(float*, int, hipStream_t);
template <int ID, int KB> void ls(float* p, int grid, hipStream_t s) { hipLaunchKernelGGL((straight<ID, KB>), dim3(grid), dim3(256), 0, s, p, 1.0001f, 0.5f); }
template <int ID> void ll(float* p, int grid, hipStream_t s, int kb) { hipLaunchKernelGGL((looped<ID>), dim3(grid), dim3(256), 0, s, p, 1.0001f, 0.5f, kb); }

template <int KB> void fill_s(std::vector<launch_t>& v) {
    v = {ls<0, KB>, ls<1, KB>, ls<2, KB>, ls<3, KB>, ls<4, KB>, ls<5, KB>, ls<6, KB>, ls<7, KB>,
         ls<8, KB>, ls<9, KB>, ls<10, KB>, ls<11, KB>, ls<12, KB>, ls<13, KB>, ls<14, KB>, ls<15, KB>};
}
static double time_seq(const std::vector<launch_t>& v, bool same, float* p, int grid, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 32; ++i) v[same ? 0 : i % 16](p, grid, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) v[same ? 0 : i % 16](p, grid, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
}
int main() {
    float* p; hipMalloc(&p, 256 * 1024 * 4); hipMemset(p, 0, 256 * 1024 * 4);
    const int reps = 1600;
    for (int grid : {256, 32}) {
        printf("grid %d workgroups of 256 threads; us per launch (same-stream, back to back)\n", grid);
        printf("  %-6s %14s %14s\n", "KB", "16 distinct", "same kernel");
        std::vector<launch_t> v;
#define ROW(KB) fill_s<KB>(v); printf("  %-6d %14.2f %14.2f\n", KB, time_seq(v, false, p, grid, reps), time_seq(v, true, p, grid, reps));
        ROW(1) ROW(2) ROW(4) ROW(8) ROW(16) ROW(32)
    }
    hipFree(p);
    return 0;
}
