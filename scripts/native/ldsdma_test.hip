// Does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept a source address that is only 8-byte aligned?  Stages 24-byte
// records (x, y, z doubles) picked by an id list into LDS as two 16-byte pieces per record -- bytes [0,16) and [8,24) -- and
// checks every value.  Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_test ldsdma_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, const int* __restrict__ ids, int n, double* out) {
    __shared__ __attribute__((aligned(16))) double A[2 * 256];   // (x, y) per slot
    __shared__ __attribute__((aligned(16))) double B[2 * 256];   // (y, z) per slot
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid < n) {
        const char* g = reinterpret_cast<const char*>(src) + 24 * (size_t)ids[tid];
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        __builtin_amdgcn_global_load_lds((gptr_t)(g), (lptr_t)(A + 2 * 64 * wave), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(g + 8), (lptr_t)(B + 2 * 64 * wave), 16, 0, 0);
    }
    __syncthreads();
    if (tid < n) { out[3 * tid] = A[2 * tid]; out[3 * tid + 1] = A[2 * tid + 1]; out[3 * tid + 2] = B[2 * tid + 1]; if (A[2 * tid + 1] != B[2 * tid]) out[3 * tid + 1] = -1e300; }
}
int main() {
    const int N = 1000, n = 200;
    std::vector<double> h(3 * N);
    for (int i = 0; i < 3 * N; ++i) h[i] = i * 1.25 + 0.5;
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = (i * 37 + 11) % N;   // odd and even ids: 24*id is 8- or 16-byte aligned
    double *d, *o; int* di;
    hipMalloc(&d, h.size() * 8); hipMalloc(&o, 3 * n * 8); hipMalloc(&di, n * 4);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(di, ids.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, di, n, o);
    std::vector<double> r(3 * n);
    if (hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    int bad = 0;
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) if (r[3 * i + c] != h[3 * ids[i] + c]) ++bad;
    std::printf("ldsdma 16-byte pieces from 8-byte aligned sources: %s (%d wrong of %d)\n", bad ? "WRONG" : "ok", bad, 3 * n);
    return bad ? 1 : 0;
}
