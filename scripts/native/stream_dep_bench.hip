// Micro-benchmark (MI355X): cost of a cross-stream dependency per round trip, two mechanisms:
//   events:      hipEventRecord(A) + hipStreamWaitEvent(B)            (what the engine uses)
//   stream ops:  hipStreamWriteValue32(A) + hipStreamWaitValue32(B)   (CP packets polling a flag in memory)
// Each round: kernel on A -> dependency -> kernel on B -> dependency -> back to A.  Build: hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int* p, int n) { int v = 0; for (int i = 0; i < n; ++i) v += i; if (v == -1) *p = v; }
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    int* d; CK(hipMalloc((void**)&d, 4));
    uint32_t* flag; CK(hipMalloc((void**)&flag, 64));
    CK(hipMemset(flag, 0, 64));
    const int R = 200;
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        uint32_t v = 0;
        for (int r = 0; r < R; ++r) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, d, 100);
            if (mode == 1) { CK(hipEventRecord(ea, a)); CK(hipStreamWaitEvent(b, ea, 0)); }
            if (mode == 2) { ++v; CK(hipStreamWriteValue32(a, flag, v, 0)); CK(hipStreamWaitValue32(b, flag, v, hipStreamWaitValueGte, 0xffffffffu)); }
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mode == 0 ? a : b, d, 100);
            if (mode == 1) { CK(hipEventRecord(eb, b)); CK(hipStreamWaitEvent(a, eb, 0)); }
            if (mode == 2) { ++v; CK(hipStreamWriteValue32(b, flag + 8, v, 0)); CK(hipStreamWaitValue32(a, flag + 8, v, hipStreamWaitValueGte, 0xffffffffu)); }
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
        std::printf("%s: %.1f us per round (2 kernels%s)\n", mode == 0 ? "one stream" : mode == 1 ? "events" : "write/wait value", us,
                    mode ? " + 2 cross-stream dependencies" : "");
    }
    return 0;
}
