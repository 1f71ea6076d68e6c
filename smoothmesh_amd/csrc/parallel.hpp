// parallel.hpp -- host threads for the one-off table builds of smgpu_create (addressing, tile tables): contiguous index
// ranges, one per thread, in order, so that per-range results can be concatenated into exactly what the serial loop builds.
// SMGPU_HOST_THREADS caps the thread count (default: the hardware's, at most 32; 1 = serial).
#pragma once
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace smgpu {

inline unsigned hostThreads() {
    static const unsigned n = [] {
        const char* e = std::getenv("SMGPU_HOST_THREADS");
        // (the default stops at 32; an explicit setting may go up to 256 -- the phases are memory-bound scatters and merges)
        if (e) return std::max(1u, std::min((unsigned)std::atoi(e), 256u));
        return std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    }();
    return n;
}

// number of ranges [0, n) is cut into (1 when n is small)
inline int rangeParts(int64_t n, int64_t grain = 1 << 15) {
    if (const char* g = std::getenv("SMGPU_HOST_GRAIN")) grain = std::atoll(g);   // (tests: force several ranges on small meshes)
    return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)hostThreads(), n / std::max<int64_t>(grain, 1)));
}

// f(part, begin, end) for parts = rangeParts(n, grain) contiguous ranges covering [0, n), each on its own thread
template <class F>
inline void parallelRanges(int64_t n, int parts, F&& f) {
    if (parts <= 1) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)parts);
    for (int t = 0; t < parts; ++t) {
        const int64_t b = n * t / parts, e = n * (t + 1) / parts;
        th.emplace_back([&f, t, b, e] { f(t, b, e); });
    }
    for (auto& x : th) x.join();
}

// the parts' vectors one after the other
template <class T>
inline void concatParts(std::vector<T>& out, std::vector<std::vector<T>>& parts) {
    size_t total = 0;
    for (auto& p : parts) total += p.size();
    out.clear();
    out.resize(total);
    std::vector<size_t> base(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); ++i) base[i + 1] = base[i] + parts[i].size();
    parallelRanges((int64_t)parts.size(), (int)parts.size(), [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            std::copy(parts[(size_t)i].begin(), parts[(size_t)i].end(), out.begin() + (ptrdiff_t)base[(size_t)i]);
            std::vector<T>().swap(parts[(size_t)i]);
        }
    });
}

// seconds since the current smgpu_create began (the verbose lines of the set-up carry it: a timeline, not just durations)
inline std::chrono::steady_clock::time_point& setupClockStart() { static std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(); return t; }
inline double setupClock() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - setupClockStart()).count(); }

// Fresh host pages of the set-up's big lists as transparent huge pages (THP mode "madvise" on the GPU boxes): the first touch of
// 4 KB pages runs at ~6 GB/s on one thread and ~20 GB/s on all of them together (the page-fault path), with 2 MB pages at 17 and
// > 100 GB/s -- and a device -> host copy into fresh pages at its pinned rate.  The advice goes to the vector's buffer BEFORE it is
// touched (reserve, advise, then resize / assign); small buffers are left alone.
inline void adviseHuge(void* p, size_t bytes) {
    const uintptr_t huge = (uintptr_t)2 << 20;
    if (!p || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + huge - 1) & ~(huge - 1), e = ((uintptr_t)p + bytes) & ~(huge - 1);
    if (e > a) (void)madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);
}
template <class V>
inline void reserveHuge(V& v, size_t n) {
    if (v.capacity() >= n) return;
    V fresh;
    fresh.reserve(n);
    adviseHuge((void*)fresh.data(), fresh.capacity() * sizeof(typename V::value_type));
    fresh.assign(v.begin(), v.end());
    v.swap(fresh);
}
template <class V>
inline void resizeHuge(V& v, size_t n) { reserveHuge(v, n); v.resize(n); }

}  // namespace smgpu
