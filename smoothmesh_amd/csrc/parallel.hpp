// parallel.hpp -- host threads for the one-off table builds of smgpu_create (addressing, tile tables): contiguous index
// ranges, one per thread, in order, so that per-range results can be concatenated into exactly what the serial loop builds.
// SMGPU_HOST_THREADS caps the thread count (default: the hardware's, at most 32; 1 = serial).
#pragma once
#include <algorithm>
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

}  // namespace smgpu
