// What does a cross-stream hand-over cost on this box?  Ping-pong of ~10 us kernels between two streams:
//   (0) both kernels on one stream (no hand-over)          (1) hipEventRecord + hipStreamWaitEvent
//   (2) hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory
// build: hipcc -O2 --offload-arch=gfx950 scripts/micro/sync_cost.hip -o build_ab/sync_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (sink && threadIdx.x == 1024) *sink = 1;
}
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    uint32_t* flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(fe)); flag = nullptr; }
    else CK(hipMemset(flag, 0, 64));
    const int N = 400;
    const unsigned long long T = 1000;   // 10 us at 100 MHz
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !flag) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t seq = 1 + rep * 4 * N;
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, T, (int*)nullptr);
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, T, (int*)nullptr);
                } else if (mode == 1) {
                    CK(hipEventRecord(e1, a));
                    CK(hipStreamWaitEvent(b, e1, 0));
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, T, (int*)nullptr);
                    CK(hipEventRecord(e2, b));
                    CK(hipStreamWaitEvent(a, e2, 0));
                } else {
                    CK(hipStreamWriteValue32(a, flag, seq, 0));
                    CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, T, (int*)nullptr);
                    ++seq;
                    CK(hipStreamWriteValue32(b, flag, seq, 0));
                    CK(hipStreamWaitValue32(a, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    ++seq;
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("mode %d rep %d: %.2f us per pair of 10-us kernels (hand-over cost per direction: %.2f us)\n", mode, rep, us, (us - 20.0) / 2);
        }
    }
    return 0;
}
