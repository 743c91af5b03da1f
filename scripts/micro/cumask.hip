// What does a CU mask on a stream (hipExtStreamCreateWithCUMask) select on this 8-XCD part, and what does an Adam sweep reach from a subset
// of the XCDs?  (Idea: the decoder weight update on a masked side stream = whole XCDs, the step's chain on the others.)
//   hipcc -O3 --offload-arch=gfx950 -o build_ab/cumask scripts/micro/cumask.hip ; ./build_ab/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

typedef float __attribute__((ext_vector_type(4))) f4;
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((((size) - 1) << 11) | ((off) << 6) | (id))

__global__ void k_where(unsigned* out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = GETREG(20, 0, 4), hw = GETREG(4, 0, 32);
        out[blockIdx.x] = (xcc << 24) | (hw & 0xFFFFFF);
    }
    // stay a little so that the grid spreads over everything the mask allows
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(4);
}
__device__ __forceinline__ float move(float p, float lm, float v) { return __builtin_fmaf(-lm, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + 1e-8f), p); }
__global__ void k_sweep(size_t n4, f4* __restrict__ W, f4* __restrict__ M, f4* __restrict__ V) {
    const size_t T = blockDim.x, per = T * 2, nb = n4 / per;
    for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
        f4 p[2], m[2], v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t e = b * per + j * T + threadIdx.x;
            p[j] = __builtin_nontemporal_load(W + e); m[j] = __builtin_nontemporal_load(M + e); v[j] = __builtin_nontemporal_load(V + e);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t e = b * per + j * T + threadIdx.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) { m[j][i] = 0.9f * m[j][i] + 1e-4f; v[j][i] = 0.999f * v[j][i] + 1e-7f; p[j][i] = move(p[j][i], 1e-4f * m[j][i], v[j][i]); }
            __builtin_nontemporal_store(p[j], W + e); __builtin_nontemporal_store(m[j], M + e); __builtin_nontemporal_store(v[j], V + e);
        }
    }
}
static void where(const char* what, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", what); return; }
    unsigned* d;
    const int G = 1024;
    hipMalloc(&d, G * 4);
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> xccs, cus;
    int per_xcc[16] = {0};
    for (unsigned x : h) { xccs.insert(x >> 24); cus.insert(((x >> 24) << 16) | ((x >> 8) & 0xFF) | (x & 0xE000)); }
    for (unsigned c : cus) per_xcc[c >> 16]++;
    printf("%-34s XCCs used:", what);
    for (int x = 0; x < 8; ++x) printf(" %d:%2d", x, per_xcc[x]);
    printf("   (%zu distinct (xcc, se, cu))\n", cus.size());
    hipFree(d);
    hipStreamDestroy(s);
}
static void sweep(const char* what, const std::vector<uint32_t>& mask, int G, size_t rows) {
    hipStream_t s;
    if (mask.empty()) hipStreamCreate(&s);
    else if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", what); return; }
    const size_t n4 = rows * 600 / 4, size = n4 * 16;
    char* base;
    hipMalloc(&base, 3 * size + (6u << 20));
    hipMemset(base, 0x3c, 3 * size + (6u << 20));
    const size_t sz2 = (size + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    f4 *W = (f4*)base, *M = (f4*)(base + sz2), *V = (f4*)(base + 2 * sz2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_sweep, dim3(G), dim3(512), 0, s, n4, W, M, V);
    float best = 1e30f;
    for (int i = 0; i < 6; ++i) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_sweep, dim3(G), dim3(512), 0, s, n4, W, M, V);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-34s %4d workgroups, %6zu rows: %.3f ms = %.2f TB/s\n", what, G, rows, best, 24.0 * rows * 600 / best / 1e9);
    hipFree(base);
    hipStreamDestroy(s);
}
static std::vector<uint32_t> bits(std::initializer_list<std::pair<int, int>> ranges) {
    std::vector<uint32_t> m(8, 0);
    for (auto r : ranges) for (int b = r.first; b < r.second; ++b) m[b / 32] |= 1u << (b % 32);
    return m;
}
int main() {
    where("all 256 bits", bits({{0, 256}}));
    where("bits 0-31", bits({{0, 32}}));
    where("bits 0-63", bits({{0, 64}}));
    where("bits 0-159", bits({{0, 160}}));
    where("bits 96-255", bits({{96, 256}}));
    {   // every bit b with b % 8 < 5
        std::vector<uint32_t> m(8, 0);
        for (int b = 0; b < 256; ++b) if (b % 8 < 5) m[b / 32] |= 1u << (b % 32);
        where("bits with b % 8 < 5", m);
        for (size_t rows : {(size_t)25024, (size_t)200000}) {
            sweep("no mask", {}, 160, rows);
            sweep("bits with b % 8 < 5 (5 XCDs?)", m, 160, rows);
            sweep("bits 0-159", bits({{0, 160}}), 160, rows);
        }
    }
    return 0;
}
