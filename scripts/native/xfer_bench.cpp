// xfer_bench.cpp -- how host <-> device copies of the set-up's lists behave on a GPU box (one-off measurement for smgpu_create's
// upload / download pipeline): pageable vs pinned vs registered memory, first touch of fresh host pages, threaded staging copies.
// build: hipcc -O2 -std=c++17 -o xfer_bench xfer_bench.cpp -pthread ; run: ./xfer_bench [MB=800]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdint>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void parCopy(char* dst, const char* src, size_t n, int T) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { const size_t b = n * t / T, e = n * (t + 1) / T; std::memcpy(dst + b, src + b, e - b); });
    for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
    const size_t n = (size_t)(argc > 1 ? std::atoi(argv[1]) : 800) << 20;
    OK(hipSetDevice(0));
    char* d = nullptr;
    double t0 = now();
    OK(hipMalloc((void**)&d, n));
    std::printf("hipMalloc %zu MB: %.1f ms\n", n >> 20, 1e3 * (now() - t0));
    char* pageable = (char*)std::malloc(n);
    t0 = now(); std::memset(pageable, 1, n); std::printf("first touch (memset, 1 thread): %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
    for (int rep = 0; rep < 2; ++rep) { t0 = now(); OK(hipMemcpy(d, pageable, n, hipMemcpyHostToDevice)); std::printf("H2D pageable: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9); }
    char* pinned = nullptr;
    t0 = now(); OK(hipHostMalloc((void**)&pinned, n, hipHostMallocDefault)); std::printf("hipHostMalloc %zu MB: %.1f ms\n", n >> 20, 1e3 * (now() - t0));
    t0 = now(); std::memset(pinned, 2, n); std::printf("memset pinned (1 thread): %.1f ms\n", 1e3 * (now() - t0));
    for (int rep = 0; rep < 2; ++rep) { t0 = now(); OK(hipMemcpy(d, pinned, n, hipMemcpyHostToDevice)); std::printf("H2D pinned: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9); }
    for (int rep = 0; rep < 2; ++rep) { t0 = now(); OK(hipMemcpy(pinned, d, n, hipMemcpyDeviceToHost)); std::printf("D2H pinned: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9); }
    for (int T : {1, 4, 8, 16, 32, 64}) { t0 = now(); parCopy(pinned, pageable, n, T); std::printf("memcpy pageable -> pinned, %2d threads: %.1f ms = %.1f GB/s\n", T, 1e3 * (now() - t0), n / (now() - t0) / 1e9); }
    {   // D2H into fresh pages
        char* fresh = (char*)std::malloc(n);
        t0 = now(); OK(hipMemcpy(fresh, d, n, hipMemcpyDeviceToHost)); std::printf("D2H pageable, fresh pages: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        t0 = now(); OK(hipMemcpy(fresh, d, n, hipMemcpyDeviceToHost)); std::printf("D2H pageable, touched pages: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        std::free(fresh);
    }
    for (int T : {8, 32}) {   // fresh pages first-touched by T threads copying out of pinned
        char* fresh = (char*)std::malloc(n);
        t0 = now(); parCopy(fresh, pinned, n, T); std::printf("memcpy pinned -> fresh pageable, %2d threads: %.1f ms = %.1f GB/s\n", T, 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        std::free(fresh);
    }
    {   // transparent huge pages for fresh host memory: madvise before the first touch
        FILE* f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char line[128] = "?";
        if (f) { if (!std::fgets(line, sizeof line, f)) line[0] = 0; std::fclose(f); }
        std::printf("transparent_hugepage/enabled: %s", line);
        char* fresh = (char*)std::malloc(n);
        const uintptr_t a = ((uintptr_t)fresh + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const int rc = madvise((void*)a, (n - (a - (uintptr_t)fresh)) & ~(size_t)((2u << 20) - 1), MADV_HUGEPAGE);
        t0 = now(); std::memset(fresh, 3, n); std::printf("first touch after madvise(MADV_HUGEPAGE) rc=%d (memset, 1 thread): %.1f ms = %.1f GB/s\n", rc, 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        std::free(fresh);
        fresh = (char*)std::malloc(n);
        const uintptr_t a2 = ((uintptr_t)fresh + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        (void)madvise((void*)a2, (n - (a2 - (uintptr_t)fresh)) & ~(size_t)((2u << 20) - 1), MADV_HUGEPAGE);
        t0 = now(); OK(hipMemcpy(fresh, d, n, hipMemcpyDeviceToHost)); std::printf("D2H pageable, fresh huge pages: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        std::free(fresh);
        fresh = (char*)std::malloc(n);
        const uintptr_t a3 = ((uintptr_t)fresh + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        (void)madvise((void*)a3, (n - (a3 - (uintptr_t)fresh)) & ~(size_t)((2u << 20) - 1), MADV_HUGEPAGE);
        t0 = now(); parCopy(fresh, pinned, n, 16); std::printf("memcpy pinned -> fresh huge pages, 16 threads: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        std::free(fresh);
    }
    {   // registering the caller's memory in place
        t0 = now(); OK(hipHostRegister(pageable, n, hipHostRegisterDefault)); std::printf("hipHostRegister %zu MB: %.1f ms\n", n >> 20, 1e3 * (now() - t0));
        t0 = now(); OK(hipMemcpy(d, pageable, n, hipMemcpyHostToDevice)); std::printf("H2D registered: %.1f ms = %.1f GB/s\n", 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        t0 = now(); OK(hipHostUnregister(pageable)); std::printf("hipHostUnregister: %.1f ms\n", 1e3 * (now() - t0));
    }
    {   // pipelined: T threads copy chunks into two pinned buffers, async H2D per chunk
        const size_t chunk = 32u << 20;
        const int nbuf = 4;
        char* stage = nullptr;
        t0 = now(); OK(hipHostMalloc((void**)&stage, chunk * nbuf, hipHostMallocDefault)); std::printf("hipHostMalloc %zu MB staging: %.1f ms\n", (chunk * nbuf) >> 20, 1e3 * (now() - t0));
        hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t ev[nbuf];
        for (auto& e : ev) OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int T : {8, 16}) {
            t0 = now();
            size_t off = 0; int i = 0;
            while (off < n) {
                const size_t len = std::min(chunk, n - off);
                char* b = stage + (size_t)(i % nbuf) * chunk;
                if (i >= nbuf) OK(hipEventSynchronize(ev[i % nbuf]));
                parCopy(b, pageable + off, len, T);
                OK(hipMemcpyAsync(d + off, b, len, hipMemcpyHostToDevice, st));
                OK(hipEventRecord(ev[i % nbuf], st));
                off += len; ++i;
            }
            OK(hipStreamSynchronize(st));
            std::printf("H2D pipelined through %d x %zu MB pinned, %2d copy threads: %.1f ms = %.1f GB/s\n", nbuf, chunk >> 20, T, 1e3 * (now() - t0), n / (now() - t0) / 1e9);
        }
    }
    return 0;
}
