// Micro-benchmark (MI355X): time per dependent kernel launch, stream launches against one hipGraph of the same chain.
// The kernels are long enough (about 40 us) that the host is never the bottleneck: what is measured is the device-side gap
// between two dependent kernels.  Build: hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = p[i];
    for (int k = 0; k < n; ++k) v = v * 1.0000001 + 1e-9;
    p[i] = v;
}
int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 4096 * 256, K = 200;
    double* d; CK(hipMalloc((void**)&d, sizeof(double) * N));
    CK(hipMemset(d, 0, sizeof(double) * N));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int iters : {2000, 4000}) {
        // one kernel alone
        float one = 0;
        hipLaunchKernelGGL(work, dim3(4096), dim3(256), 0, s, d, iters);
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(work, dim3(4096), dim3(256), 0, s, d, iters);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&one, e0, e1));
        // chain on the stream
        float chain = 0;
        CK(hipEventRecord(e0, s));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(work, dim3(4096), dim3(256), 0, s, d, iters);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&chain, e0, e1));
        // the same chain as a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(work, dim3(4096), dim3(256), 0, s, d, iters);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float gr = 0;
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&gr, e0, e1));
        std::printf("kernel %.1f us: stream chain %.2f us per launch, graph %.2f us per launch\n", 1e3 * one, 1e3 * chain / K, 1e3 * gr / K);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
