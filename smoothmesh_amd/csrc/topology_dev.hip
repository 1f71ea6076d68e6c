// topology_dev.hip -- Topology::build (topology.cpp) on the device: the same lists in the same orders, as sort / scan / fill
// passes instead of host loops.  smgpu_create spent 3.5 of its ~5 s for 10 M cells in the host build, a serial CHAIN of ten
// memory-bound phases that host threads do not shorten; here every list is one radix sort of 64-bit (row, value)
// keys (rocPRIM) plus a few streaming kernels:
//   cellFacesGeom  (cell, side, face)            pointFaces    (point, position in the face -> point list)
//   pointCells     (point, cell), unique         edges         (lo, hi) per face edge, unique; the rank is the edge id
//   pointEdges     (point, edge)                 edgeFaces     (edge, face)
// and one thread per edge for what the reference derives per edge (edgeCells in first-appearance order with the cell's two edge
// faces -- findCellFacePair SM.C:1042-1097 --, the ring order).  Row orders are the host build's by construction: a row is its keys
// in ascending order, and the host fills ascending too (tests/test_gpu_topology.py compares every array of both builds).
// What the device path does not handle it hands back (return 1): meshes on which the host build reports an error (it words the
// message), edges with more than kMaxEdgeFaces faces, points with more than 255 neighbours.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "topology.hpp"
#include "parallel.hpp"

namespace smgpu {

namespace {

constexpr int kTB = 256;
constexpr int kMaxEdgeFaces = 16;      // faces around an edge the per-edge kernels keep in registers (more: host build)
typedef unsigned long long u64;

#define TD_OK(expr)                                                                                             \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) { why = std::string(#expr) + ": " + hipGetErrorString(e__); return 2; }          \
    } while (0)

inline int gridOf(int64_t n) { return (int)std::max<int64_t>(1, (n + kTB - 1) / kTB); }

struct Flags { int bad; int maxFace; int maxEdgeFaces; int maxPointCells; int maxPointPoints; int nEdges; int nPointCells; int pad; };

// ---- kernels -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTB) k_td_faces(int nF, int nIF, int nP, int nC, const int* __restrict__ fo, const int* __restrict__ fp, const int* __restrict__ own,
                                                   const int* __restrict__ nei, int* __restrict__ faceOf, Flags* fl) {
    const int f = blockIdx.x * kTB + threadIdx.x;
    if (f >= nF) return;
    const int b = fo[f], n = fo[f + 1] - b;
    bool bad = n < 3 || own[f] < 0 || own[f] >= nC || (f < nIF && (nei[f] < 0 || nei[f] >= nC));
    for (int i = 0; i < n; ++i) {
        const int a = fp[b + i], c = fp[b + (i == n - 1 ? 0 : i + 1)];
        if (a < 0 || a >= nP || a == c) bad = true;      // (label out of range; degenerate edge)
        faceOf[b + i] = f;
    }
    if (bad) fl->bad = 1;
    atomicMax(&fl->maxFace, n);
}
// cell -> faces keys: (cell, side, face); owned faces first (side 0), then the neighboured ones
__global__ void __launch_bounds__(kTB) k_td_cfKeys(int nF, int nIF, const int* __restrict__ own, const int* __restrict__ nei, u64* __restrict__ keys) {
    const int i = blockIdx.x * kTB + threadIdx.x;
    if (i < nF) keys[i] = ((u64)(unsigned)own[i] << 32) | (unsigned)i;
    else if (i < nF + nIF) { const int f = i - nF; keys[i] = ((u64)(unsigned)nei[f] << 32) | 0x80000000u | (unsigned)f; }
}
__global__ void __launch_bounds__(kTB) k_td_low32(const u64* __restrict__ keys, int64_t n, int* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) out[i] = (int)(unsigned)(keys[i] & 0xffffffffull);
}
// off[r] = first index whose key's row (high word) is >= r, r = 0 .. rows
__global__ void __launch_bounds__(kTB) k_td_offsets(const u64* __restrict__ keys, int64_t n, int rows, int* __restrict__ off) {
    const int r = blockIdx.x * kTB + threadIdx.x;
    if (r > rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((keys[mid] >> 32) < (u64)(unsigned)r) lo = mid + 1; else hi = mid; }
    off[r] = (int)lo;
}
__global__ void __launch_bounds__(kTB) k_td_pfKeys(int64_t nnz, const int* __restrict__ fp, u64* __restrict__ keys) {
    const int64_t k = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (k < nnz) keys[k] = ((u64)(unsigned)fp[k] << 32) | (unsigned)k;
}
__global__ void __launch_bounds__(kTB) k_td_pfFill(int64_t nnz, const u64* __restrict__ keys, const int* __restrict__ fo, const int* __restrict__ fp, const int* __restrict__ faceOf,
                                                    int* __restrict__ pfFace, int* __restrict__ pfPrev, int* __restrict__ pfNext) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= nnz) return;
    const int k = (int)(unsigned)(keys[j] & 0xffffffffull);
    const int f = faceOf[k], b = fo[f], n = fo[f + 1] - b, i = k - b;
    pfFace[j] = f;
    pfPrev[j] = fp[b + (i == 0 ? n - 1 : i - 1)];
    pfNext[j] = fp[b + (i == n - 1 ? 0 : i + 1)];
}
// (point, cell) for the owner and, on internal faces, the neighbour of every face the point is in; ~0 where there is none
__global__ void __launch_bounds__(kTB) k_td_pcKeys(int64_t nnz, int nIF, const int* __restrict__ fp, const int* __restrict__ faceOf, const int* __restrict__ own,
                                                    const int* __restrict__ nei, u64* __restrict__ keys) {
    const int64_t k = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (k >= nnz) return;
    const int f = faceOf[k];
    const u64 p = (u64)(unsigned)fp[k] << 32;
    keys[2 * k] = p | (unsigned)own[f];
    keys[2 * k + 1] = (f < nIF) ? (p | (unsigned)nei[f]) : ~0ull;
}
struct KeyEq { __device__ bool operator()(const u64& a, const u64& b) const { return a == b; } };
__global__ void __launch_bounds__(kTB) k_td_edgeKeys(int64_t nnz, const int* __restrict__ fo, const int* __restrict__ fp, const int* __restrict__ faceOf, u64* __restrict__ keys,
                                                      int* __restrict__ vals) {
    const int64_t k = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (k >= nnz) return;
    const int f = faceOf[k], b = fo[f], n = fo[f + 1] - b, i = (int)k - b;
    const int a = fp[k], c = fp[b + (i == n - 1 ? 0 : i + 1)];
    keys[k] = ((u64)(unsigned)min(a, c) << 32) | (unsigned)max(a, c);
    vals[k] = (int)k;
}
// heads[j] = 1 where a new key starts
__global__ void __launch_bounds__(kTB) k_td_heads(int64_t n, const u64* __restrict__ keys, int* __restrict__ heads) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j < n) heads[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}
// rank[j] = inclusive scan of heads: edge id = rank - 1; the unique keys are the edges, every face edge learns its edge id
__global__ void __launch_bounds__(kTB) k_td_edgeScatter(int64_t n, const u64* __restrict__ keys, const int* __restrict__ vals, const int* __restrict__ rank, int* __restrict__ edges,
                                                         int* __restrict__ faceEdge) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const int e = rank[j] - 1;
    faceEdge[vals[j]] = e;
    if (j == 0 || keys[j] != keys[j - 1]) { edges[2 * (size_t)e] = (int)(keys[j] >> 32); edges[2 * (size_t)e + 1] = (int)(unsigned)(keys[j] & 0xffffffffull); }
}
__global__ void __launch_bounds__(kTB) k_td_peKeys(int nE, const int* __restrict__ edges, u64* __restrict__ keys) {
    const int e = blockIdx.x * kTB + threadIdx.x;
    if (e >= nE) return;
    keys[2 * (size_t)e] = ((u64)(unsigned)edges[2 * e] << 32) | (unsigned)e;
    keys[2 * (size_t)e + 1] = ((u64)(unsigned)edges[2 * e + 1] << 32) | (unsigned)e;
}
__global__ void __launch_bounds__(kTB) k_td_peFill(int64_t n, const u64* __restrict__ keys, const int* __restrict__ edges, int* __restrict__ peEdge, int* __restrict__ ppPt) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const int p = (int)(keys[j] >> 32), e = (int)(unsigned)(keys[j] & 0xffffffffull);
    peEdge[j] = e;
    ppPt[j] = (edges[2 * e] == p) ? edges[2 * e + 1] : edges[2 * e];
}
__global__ void __launch_bounds__(kTB) k_td_rowMax(int rows, const int* __restrict__ off, int* out) {
    const int r = blockIdx.x * kTB + threadIdx.x;
    int v = (r < rows) ? off[r + 1] - off[r] : 0;
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
    // (the word only grows: a wave whose maximum it already holds skips the atomic -- all but a few do; 470 k atomics on one word
    // took 3 ms)
    if ((threadIdx.x & 63) == 0 && v > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, v);
}
// prev / next vertex of every pointFaces entry as a slot of the point's pointPoints row (the last match below 255, as the host loop)
__global__ void __launch_bounds__(kTB) k_td_pfSlots(int nP, const int* __restrict__ pfOff, const int* __restrict__ pfPrev, const int* __restrict__ pfNext, const int* __restrict__ ppOff,
                                                     const int* __restrict__ ppPt, uint8_t* __restrict__ prevSlot, uint8_t* __restrict__ nextSlot) {
    const int p = blockIdx.x * kTB + threadIdx.x;
    if (p >= nP) return;
    const int nb = ppOff[p], nv = min(ppOff[p + 1] - nb, 255);
    for (int k = pfOff[p]; k < pfOff[p + 1]; ++k) {
        int a = 255, c = 255;
        const int pv = pfPrev[k], nx = pfNext[k];
        for (int j = 0; j < nv; ++j) {
            const int q = ppPt[nb + j];
            if (q == pv) a = j;
            if (q == nx) c = j;
        }
        prevSlot[k] = (uint8_t)a; nextSlot[k] = (uint8_t)c;
    }
}
__global__ void __launch_bounds__(kTB) k_td_efKeys(int64_t nnz, const int* __restrict__ faceEdge, const int* __restrict__ faceOf, u64* __restrict__ keys) {
    const int64_t k = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (k < nnz) keys[k] = ((u64)(unsigned)faceEdge[k] << 32) | (unsigned)faceOf[k];
}
// the cells of one edge in first-appearance order through its (ascending) faces, owner then neighbour, with the cell's two edge
// faces (topology.cpp; findCellFacePair SM.C:1042-1097).  COUNT: only the number of cells; else the row at ecOff[e].
template <bool COUNT>
__global__ void __launch_bounds__(kTB) k_td_edgeCells(int nE, int nIF, const int* __restrict__ efOff, const int* __restrict__ efFace, const int* __restrict__ own,
                                                       const int* __restrict__ nei, int* __restrict__ cnt, const int* __restrict__ ecOff, int* __restrict__ ecCell,
                                                       uint8_t* __restrict__ ecF0, uint8_t* __restrict__ ecF1, Flags* fl) {
    const int e = blockIdx.x * kTB + threadIdx.x;
    if (e >= nE) return;
    const int b = efOff[e], n = efOff[e + 1] - b;
    if (n > kMaxEdgeFaces) { fl->bad = 1; if (COUNT) cnt[e] = 0; return; }
    int fo[kMaxEdgeFaces], fn[kMaxEdgeFaces];
    int cells[2 * kMaxEdgeFaces];
    int nc = 0;
#pragma unroll
    for (int i = 0; i < kMaxEdgeFaces; ++i) {
        if (i < n) {
            const int f = efFace[b + i];
            fo[i] = own[f]; fn[i] = (f < nIF) ? nei[f] : -1;
        } else { fo[i] = -2; fn[i] = -2; }
    }
    for (int i = 0; i < n; ++i) {
        bool seen = false;
        for (int q = 0; q < nc; ++q) seen = seen || cells[q] == fo[i];
        if (!seen) cells[nc++] = fo[i];
        if (fn[i] >= 0) {
            seen = false;
            for (int q = 0; q < nc; ++q) seen = seen || cells[q] == fn[i];
            if (!seen) cells[nc++] = fn[i];
        }
    }
    if (COUNT) { cnt[e] = nc; return; }
    const int o = ecOff[e];
    for (int q = 0; q < nc; ++q) {
        const int c = cells[q];
        int f0 = -1, f1 = -1, hits = 0;
        for (int i = 0; i < n; ++i)
            if (fo[i] == c || fn[i] == c) { if (hits == 0) f0 = i; else if (hits == 1) f1 = i; ++hits; }
        if (hits != 2) fl->bad = 1;      // (the host build words the reference's "Sanity broken" messages)
        ecCell[o + q] = c; ecF0[o + q] = (uint8_t)f0; ecF1[o + q] = (uint8_t)f1;
    }
}
// ring order of the faces / cells around one edge (topology.cpp, "ring order around each edge")
__global__ void __launch_bounds__(kTB) k_td_rings(int nE, const int* __restrict__ efOff, const int* __restrict__ efFace, const int* __restrict__ ecOff, const int* __restrict__ ecCell,
                                                   const uint8_t* __restrict__ ecF0, const uint8_t* __restrict__ ecF1, int* __restrict__ ringFace, int* __restrict__ ringCell,
                                                   uint8_t* __restrict__ ringOk) {
    const int e = blockIdx.x * kTB + threadIdx.x;
    if (e >= nE) return;
    const int fb = efOff[e], nf = efOff[e + 1] - fb, cb = ecOff[e], nc = ecOff[e + 1] - cb;
    for (int i = 0; i < nf; ++i) ringFace[fb + i] = -1;
    for (int i = 0; i < nc; ++i) ringCell[cb + i] = -1;
    ringOk[e] = 0;
    if (nc < 1 || (nc != nf && nc != nf - 1) || nf > kMaxEdgeFaces) return;
    int deg[kMaxEdgeFaces], c0[kMaxEdgeFaces], c1[kMaxEdgeFaces];
    unsigned used = 0u;
#pragma unroll
    for (int i = 0; i < kMaxEdgeFaces; ++i) { deg[i] = 0; c0[i] = (i < nc) ? ecF0[cb + i] : 255; c1[i] = (i < nc) ? ecF1[cb + i] : 255; }
    for (int i = 0; i < nc; ++i)
#pragma unroll
        for (int q = 0; q < kMaxEdgeFaces; ++q) { if (q == c0[i]) deg[q]++; if (q == c1[i]) deg[q]++; }
    bool ok = true;
    int start = 0, nEnds = 0;
#pragma unroll
    for (int i = 0; i < kMaxEdgeFaces; ++i)
        if (i < nf) {
            if (deg[i] == 1) { if (nEnds == 0) start = i; ++nEnds; }
            else if (deg[i] != 2) ok = false;
        }
    if (!ok || (nc == nf && nEnds != 0) || (nc == nf - 1 && nEnds != 2)) return;
    int cur = start, placed = 0;
    ringFace[fb] = efFace[fb + cur];
    while (placed < nc) {
        int next = -1, ci = -1;
        for (int i = 0; i < nc; ++i) {
            if ((used >> i) & 1u) continue;
            if (c0[i] == cur) { next = c1[i]; ci = i; break; }
            if (c1[i] == cur) { next = c0[i]; ci = i; break; }
        }
        if (ci < 0) break;
        used |= 1u << ci;
        ringCell[cb + placed] = ecCell[cb + ci];
        ++placed;
        if (placed < nf) ringFace[fb + placed] = efFace[fb + next];
        cur = next;
    }
    if (placed == nc && (nc == nf - 1 || cur == start)) ringOk[e] = 1;
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct DevBuf {
    std::vector<void*> all;
    ~DevBuf() { for (void* p : all) (void)hipFree(p); }
    template <class T> T* get(size_t n, std::string& why) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); why = "device allocation failed"; return nullptr; }
        all.push_back(p);
        return (T*)p;
    }
    void drop(void* p) { auto it = std::find(all.begin(), all.end(), p); if (it != all.end()) { (void)hipFree(p); all.erase(it); } }
    void release(void* p) { auto it = std::find(all.begin(), all.end(), p); if (it != all.end()) all.erase(it); }      // the caller owns it now
};

// Host -> device copies of the caller's (pageable) lists: chunks are copied into pinned staging buffers by several threads and sent
// from there, the next chunk being filled while the last one travels.  A plain hipMemcpyAsync from pageable memory went at
// 4.7 GB/s beside the set-up's other host threads (0.18 s for the 856 MB of the 10 M-cell mesh); the staged form runs at the
// link's ~53 GB/s (scripts/native/xfer_bench.cpp).
struct StagedUpload {
    static constexpr size_t kChunk = (size_t)32 << 20;
    static constexpr int kBufs = 4, kThreads = 8;
    char* stage = nullptr;
    hipEvent_t ev[kBufs] = {};
    int turn = 0;
    bool ok = false;
    double tFill = 0.0, tWait = 0.0, tAlloc = 0.0;      // (SMGPU_VERBOSE >= 2: where an upload's time went)
    static double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    StagedUpload() {
        const double t0 = nowS();
        struct Done { double& t; double t0; ~Done() { t = nowS() - t0; } } done{tAlloc, t0};
        if (hipHostMalloc((void**)&stage, kChunk * kBufs, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); stage = nullptr; return; }
        for (auto& e : ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        ok = true;
    }
    ~StagedUpload() {
        for (auto& e : ev) if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
        if (stage) (void)hipHostFree(stage);
    }
    hipError_t copy(void* dst, const void* src, size_t bytes, hipStream_t st) {
        if (!ok || bytes < kChunk / 4) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
        for (size_t off = 0; off < bytes; off += kChunk, ++turn) {
            const size_t len = std::min(kChunk, bytes - off);
            char* b = stage + (size_t)(turn % kBufs) * kChunk;
            double t0 = nowS();
            if (turn >= kBufs) { const hipError_t e = hipEventSynchronize(ev[turn % kBufs]); if (e != hipSuccess) return e; }
            tWait += nowS() - t0;
            t0 = nowS();
            std::vector<std::thread> th;
            const char* from = (const char*)src + off;
            for (int t = 1; t < kThreads; ++t) th.emplace_back([=] { const size_t lo = len * t / kThreads, hi = len * (t + 1) / kThreads; std::memcpy(b + lo, from + lo, hi - lo); });
            std::memcpy(b, from, len / kThreads);
            for (auto& x : th) x.join();
            tFill += nowS() - t0;
            hipError_t e = hipMemcpyAsync((char*)dst + off, b, len, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipEventRecord(ev[turn % kBufs], st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
};

int bitsFor(int64_t rows) { int b = 1; while (((int64_t)1 << b) < rows + 1 && b < 31) ++b; return b; }

// device -> host copies of whole lists: the arrays are sized by one thread each (first touch of fresh pages) and copied in 32 MB
// pieces by four threads with a stream each (one pageable copy at a time ran at ~1.1 GB/s: 1.4 s of the 1.65 s the build took
// for 10 M cells)
struct DownJob { std::function<void*()> size; const void* src; size_t bytes; void* dst; };
template <class Vec, class T> void wantDown(std::vector<DownJob>& jobs, Vec& vec, const T* src, size_t n) {
    Vec* v = &vec;
    jobs.push_back(DownJob{[v, n]() -> void* { resizeHuge(*v, n); return (void*)v->data(); }, (const void*)src, n * sizeof(T), nullptr});
}
bool runDown(std::vector<DownJob>& jobs, int device) {
    {   // sizes first (a thread per array), then the copies
        std::vector<std::thread> th;
        for (DownJob& j : jobs) th.emplace_back([&j] { j.dst = j.size(); });
        for (auto& x : th) x.join();
    }
    struct Piece { char* dst; const char* src; size_t bytes; };
    std::vector<Piece> pieces;
    const size_t chunk = (size_t)32 << 20;
    for (const DownJob& j : jobs)
        for (size_t o = 0; o < j.bytes; o += chunk) pieces.push_back(Piece{(char*)j.dst + o, (const char*)j.src + o, std::min(chunk, j.bytes - o)});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    for (int w = 0; w < 4; ++w)
        th.emplace_back([&] {
            if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
            hipStream_t s2 = nullptr;
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { failed = 1; return; }
            for (size_t i = next++; i < pieces.size(); i = next++)
                if (hipMemcpyAsync(pieces[i].dst, pieces[i].src, pieces[i].bytes, hipMemcpyDeviceToHost, s2) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess) failed = 1;
            (void)hipStreamDestroy(s2);
        });
    for (auto& x : th) x.join();
    jobs.clear();
    return failed == 0;
}

}  // namespace

// 0: t holds the addressing (device build); 1: not handled here (the caller runs the host build); 2: a HIP error (why)
int buildTopologyOnDevice(Topology& t, int32_t nP, int32_t nC, int32_t nF, int32_t nIF, const int32_t* faceOffsets, const int32_t* facePts, const int32_t* own,
                          const int32_t* nei, int device, std::string& why, const std::function<void()>& afterCells, const std::function<void()>& afterPoints,
                          DeviceTopologyArrays* keep, const std::function<void()>& afterEdges, bool deferUnread) {
    const bool verbose = std::getenv("SMGPU_VERBOSE") && std::atoi(std::getenv("SMGPU_VERBOSE")) >= 2;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        static const bool sync = std::atoi(std::getenv("SMGPU_VERBOSE")) >= 3;      // (3: every stage waited for -- its own duration, not the pipeline's)
        if (sync) (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[smgpu] device addressing: %-20s %.3f s   (%s at +%.3f s)\n", what, std::chrono::duration<double>(now - t0).count(), sync ? "done" : "enqueued", setupClock());
        t0 = now;
    };
    if (nP <= 0 || nC <= 0 || nF <= 0 || nIF < 0 || nIF > nF) return 1;
    const int64_t nnz = faceOffsets[nF];
    if (nnz >= ((int64_t)1 << 30) || nnz < 3) return 1;      // (2 * nnz keys must index with int32)
    TD_OK(hipSetDevice(device));
    lap("HIP context");
    // Host pages are touched ahead of the copies that fill them: the copies of the caller's lists and the sizing of the lists to be
    // downloaded (first touch of 2 GB of fresh pages for 10 M cells) run on threads of their own beside the uploads, the kernels
    // and the earlier stages of the download -- they were 0.2 s of the download's critical path.  (Joined before any return.)
    struct Background {
        std::vector<std::thread> th;
        void join() { for (auto& x : th) if (x.joinable()) x.join(); th.clear(); }
        ~Background() { join(); }
    } bg;
    bg.th.emplace_back([&t, faceOffsets, own, nei, nF, nIF] {
        reserveHuge(t.facePoints.off, (size_t)nF + 1); t.facePoints.off.assign(faceOffsets, faceOffsets + nF + 1);
        reserveHuge(t.owner, (size_t)nF); t.owner.assign(own, own + nF);
        reserveHuge(t.neighbour, (size_t)nIF); t.neighbour.assign(nei, nei + nIF); });
    bg.th.emplace_back([&t, facePts, nnz] { reserveHuge(t.facePoints.val, (size_t)nnz); t.facePoints.val.assign(facePts, facePts + nnz); });
    bg.th.emplace_back([&t, nC, nF, nIF] { resizeHuge(t.cellFacesGeom.off, (size_t)nC + 1); resizeHuge(t.cellFacesGeom.val, (size_t)nF + (size_t)nIF); });
    hipStream_t st = nullptr;
    DevBuf D;
    Flags* fl = D.get<Flags>(1, why);
    int *dFo = D.get<int>((size_t)nF + 1, why), *dFp = D.get<int>((size_t)nnz, why), *dOwn = D.get<int>((size_t)nF, why), *dNei = D.get<int>((size_t)nIF, why);
    int* faceOf = D.get<int>((size_t)nnz, why);
    const size_t maxKeys = (size_t)std::max<int64_t>(2 * nnz, (int64_t)nF + nIF);
    u64 *kA = D.get<u64>(maxKeys, why), *kB = D.get<u64>(maxKeys, why);
    int *vA = D.get<int>((size_t)nnz, why), *vB = D.get<int>((size_t)nnz, why);
    if (!fl || !dFo || !dFp || !dOwn || !dNei || !faceOf || !kA || !kB || !vA || !vB) return why.empty() ? 1 : 2;
    lap("device buffers");
    TD_OK(hipMemsetAsync(fl, 0, sizeof(Flags), st));
    {
        std::unique_ptr<StagedUpload> up(new StagedUpload());
        TD_OK(up->copy(dFo, faceOffsets, ((size_t)nF + 1) * 4, st));
        TD_OK(up->copy(dFp, facePts, (size_t)nnz * 4, st));
        TD_OK(up->copy(dOwn, own, (size_t)nF * 4, st));
        if (nIF) TD_OK(up->copy(dNei, nei, (size_t)nIF * 4, st));
        if (verbose) std::fprintf(stderr, "[smgpu] staged upload: staging buffers %.3f s, filling them %.3f s, waiting for the link %.3f s\n", up->tAlloc, up->tFill, up->tWait);
        // (releasing pinned memory waits for the copies and costs as much as pinning it: on a thread of its own, joined with the others)
        StagedUpload* u = up.release();
        bg.th.emplace_back([u, verbose] { const double t0 = StagedUpload::nowS(); delete u; if (verbose) std::fprintf(stderr, "[smgpu] staged upload: buffers released in %.3f s\n", StagedUpload::nowS() - t0); });
    }
    lap("upload");
    // rocPRIM temporary storage: sized for the largest sort / scan of the build
    size_t tempBytes = 0;
    {
        size_t b = 0;
        (void)rocprim::radix_sort_keys(nullptr, b, kA, kB, maxKeys, 0, 64, st); tempBytes = std::max(tempBytes, b);
        (void)rocprim::radix_sort_pairs(nullptr, b, kA, kB, vA, vB, (size_t)nnz, 0, 64, st); tempBytes = std::max(tempBytes, b);
        (void)rocprim::inclusive_scan(nullptr, b, vA, vB, (size_t)std::max<int64_t>(nnz, 1), rocprim::plus<int>(), st); tempBytes = std::max(tempBytes, b);
        (void)rocprim::unique(nullptr, b, kA, kB, (int*)nullptr, maxKeys, KeyEq(), st); tempBytes = std::max(tempBytes, b);
    }
    void* temp = D.get<char>(tempBytes + 256, why);
    if (!temp) return 2;
    auto sortKeys = [&](u64* in, u64* out, size_t n, int rowBits) -> hipError_t {
        size_t b = tempBytes;
        return rocprim::radix_sort_keys(temp, b, in, out, n, 0, (unsigned)std::min(64, 32 + rowBits), st);
    };
    hipLaunchKernelGGL(k_td_faces, dim3(gridOf(nF)), dim3(kTB), 0, st, nF, nIF, nP, nC, dFo, dFp, dOwn, dNei, faceOf, fl);
    lap("faces");

    // ---- cell -> faces (geometry accumulation order) ----
    const int64_t nCF = (int64_t)nF + nIF;
    int *cfOff = D.get<int>((size_t)nC + 1, why), *cfVal = D.get<int>((size_t)nCF, why);
    if (!cfOff || !cfVal) return 2;
    hipLaunchKernelGGL(k_td_cfKeys, dim3(gridOf(nCF)), dim3(kTB), 0, st, nF, nIF, dOwn, dNei, kA);
    TD_OK(sortKeys(kA, kB, (size_t)nCF, bitsFor(nC)));
    hipLaunchKernelGGL(k_td_low32, dim3(gridOf(nCF)), dim3(kTB), 0, st, kB, nCF, cfVal);
    hipLaunchKernelGGL(k_td_offsets, dim3(gridOf((int64_t)nC + 1)), dim3(kTB), 0, st, kB, nCF, nC, cfOff);
    lap("cellFaces");

    // ---- pointFaces with prev / next vertex ----
    int *pfOff = D.get<int>((size_t)nP + 1, why), *pfFace = D.get<int>((size_t)nnz, why), *pfPrev = D.get<int>((size_t)nnz, why), *pfNext = D.get<int>((size_t)nnz, why);
    if (!pfOff || !pfFace || !pfPrev || !pfNext) return 2;
    hipLaunchKernelGGL(k_td_pfKeys, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, dFp, kA);
    TD_OK(sortKeys(kA, kB, (size_t)nnz, bitsFor(nP)));
    hipLaunchKernelGGL(k_td_pfFill, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, kB, dFo, dFp, faceOf, pfFace, pfPrev, pfNext);
    hipLaunchKernelGGL(k_td_offsets, dim3(gridOf((int64_t)nP + 1)), dim3(kTB), 0, st, kB, nnz, nP, pfOff);
    lap("pointFaces");

    // ---- pointCells: (point, cell) sorted, unique ----
    int* pcOff = D.get<int>((size_t)nP + 1, why);
    int* dCount = D.get<int>(4, why);
    if (!pcOff || !dCount) return 2;
    hipLaunchKernelGGL(k_td_pcKeys, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, nIF, dFp, faceOf, dOwn, dNei, kA);
    TD_OK(sortKeys(kA, kB, (size_t)(2 * nnz), bitsFor(nP) + 1));      // (+1: the ~0 keys sort behind every point)
    { size_t b = tempBytes; TD_OK(rocprim::unique(temp, b, kB, kA, dCount, (size_t)(2 * nnz), KeyEq(), st)); }
    int nUnique = 0;
    TD_OK(hipMemcpyAsync(&nUnique, dCount, 4, hipMemcpyDeviceToHost, st));
    TD_OK(hipStreamSynchronize(st));
    u64 lastKey = 0;
    if (nUnique > 0) TD_OK(hipMemcpy(&lastKey, kA + (nUnique - 1), 8, hipMemcpyDeviceToHost));
    const int64_t nPC = nUnique - ((nUnique > 0 && lastKey == ~0ull) ? 1 : 0);
    int* pcVal = D.get<int>((size_t)nPC, why);
    if (!pcVal) return 2;
    hipLaunchKernelGGL(k_td_low32, dim3(gridOf(nPC)), dim3(kTB), 0, st, kA, nPC, pcVal);
    hipLaunchKernelGGL(k_td_offsets, dim3(gridOf((int64_t)nP + 1)), dim3(kTB), 0, st, kA, nPC, nP, pcOff);
    hipLaunchKernelGGL(k_td_rowMax, dim3(gridOf(nP)), dim3(kTB), 0, st, nP, pcOff, &fl->maxPointCells);
    lap("pointCells");

    // ---- edges: (lo, hi) of every face edge, sorted; the unique keys in order are the edges ----
    int* faceEdge = D.get<int>((size_t)nnz, why);
    if (!faceEdge) return 2;
    hipLaunchKernelGGL(k_td_edgeKeys, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, dFo, dFp, faceOf, kA, vA);
    { size_t b = tempBytes; TD_OK(rocprim::radix_sort_pairs(temp, b, kA, kB, vA, vB, (size_t)nnz, 0, (unsigned)std::min(64, 32 + bitsFor(nP)), st)); }
    int* heads = vA;      // (free again)
    hipLaunchKernelGGL(k_td_heads, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, kB, heads);
    int* rank = D.get<int>((size_t)nnz, why);
    if (!rank) return 2;
    { size_t b = tempBytes; TD_OK(rocprim::inclusive_scan(temp, b, heads, rank, (size_t)nnz, rocprim::plus<int>(), st)); }
    int nE = 0;
    TD_OK(hipMemcpyAsync(&nE, rank + (nnz - 1), 4, hipMemcpyDeviceToHost, st));
    TD_OK(hipStreamSynchronize(st));
    int* edges = D.get<int>(2 * (size_t)nE, why);
    if (!edges) return 2;
    {   // the sizes of the edge and point lists are known from here on
        const size_t e2 = 2 * (size_t)nE, ne1 = (size_t)nE + 1, np1 = (size_t)nP + 1, npc = (size_t)nPC, nz = (size_t)nnz;
        bg.th.emplace_back([&t, e2, ne1] { resizeHuge(t.edges, e2); resizeHuge(t.edgeFaces.off, ne1); resizeHuge(t.edgeCells.off, ne1); });
        bg.th.emplace_back([&t, nz] { resizeHuge(t.edgeFaces.val, nz); });
        bg.th.emplace_back([&t, e2, np1] { resizeHuge(t.pointPoints, e2); resizeHuge(t.pointEdges.off, np1); resizeHuge(t.pointCells.off, np1); });
        bg.th.emplace_back([&t, npc] { resizeHuge(t.pointCells.val, npc); });
    }
    hipLaunchKernelGGL(k_td_edgeScatter, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, kB, vB, rank, edges, faceEdge);
    D.drop(rank);
    lap("edges");

    // ---- pointEdges / pointPoints ----
    int *ppOff = D.get<int>((size_t)nP + 1, why), *peEdge = D.get<int>(2 * (size_t)nE, why), *ppPt = D.get<int>(2 * (size_t)nE, why);
    if (!ppOff || !peEdge || !ppPt) return 2;
    hipLaunchKernelGGL(k_td_peKeys, dim3(gridOf(nE)), dim3(kTB), 0, st, nE, edges, kA);
    TD_OK(sortKeys(kA, kB, 2 * (size_t)nE, bitsFor(nP)));
    hipLaunchKernelGGL(k_td_peFill, dim3(gridOf(2 * (int64_t)nE)), dim3(kTB), 0, st, 2 * (int64_t)nE, kB, edges, peEdge, ppPt);
    hipLaunchKernelGGL(k_td_offsets, dim3(gridOf((int64_t)nP + 1)), dim3(kTB), 0, st, kB, 2 * (int64_t)nE, nP, ppOff);
    hipLaunchKernelGGL(k_td_rowMax, dim3(gridOf(nP)), dim3(kTB), 0, st, nP, ppOff, &fl->maxPointPoints);
    uint8_t *prevSlot = D.get<uint8_t>((size_t)nnz, why), *nextSlot = D.get<uint8_t>((size_t)nnz, why);
    if (!prevSlot || !nextSlot) return 2;
    hipLaunchKernelGGL(k_td_pfSlots, dim3(gridOf(nP)), dim3(kTB), 0, st, nP, pfOff, pfPrev, pfNext, ppOff, ppPt, prevSlot, nextSlot);
    lap("pointEdges + slots");

    // ---- edgeFaces (ascending face id; a face that holds the edge twice is listed twice, as the host fill does) ----
    int *efOff = D.get<int>((size_t)nE + 1, why), *efFace = D.get<int>((size_t)nnz, why);
    if (!efOff || !efFace) return 2;
    hipLaunchKernelGGL(k_td_efKeys, dim3(gridOf(nnz)), dim3(kTB), 0, st, nnz, faceEdge, faceOf, kA);
    TD_OK(sortKeys(kA, kB, (size_t)nnz, bitsFor(nE)));
    hipLaunchKernelGGL(k_td_low32, dim3(gridOf(nnz)), dim3(kTB), 0, st, kB, nnz, efFace);
    hipLaunchKernelGGL(k_td_offsets, dim3(gridOf((int64_t)nE + 1)), dim3(kTB), 0, st, kB, nnz, nE, efOff);
    hipLaunchKernelGGL(k_td_rowMax, dim3(gridOf(nE)), dim3(kTB), 0, st, nE, efOff, &fl->maxEdgeFaces);
    lap("edgeFaces");

    // ---- edgeCells + face pairs, rings ----
    int *ecCnt = D.get<int>((size_t)nE + 1, why), *ecOff = D.get<int>((size_t)nE + 1, why);
    if (!ecCnt || !ecOff) return 2;
    hipLaunchKernelGGL(k_td_edgeCells<true>, dim3(gridOf(nE)), dim3(kTB), 0, st, nE, nIF, efOff, efFace, dOwn, dNei, ecCnt, (const int*)nullptr, (int*)nullptr,
                       (uint8_t*)nullptr, (uint8_t*)nullptr, fl);
    TD_OK(hipMemsetAsync(ecOff, 0, 4, st));
    { size_t b = tempBytes; TD_OK(rocprim::inclusive_scan(temp, b, ecCnt, ecOff + 1, (size_t)nE, rocprim::plus<int>(), st)); }
    int nEC = 0;
    TD_OK(hipMemcpyAsync(&nEC, ecOff + nE, 4, hipMemcpyDeviceToHost, st));
    Flags hf{};
    TD_OK(hipMemcpyAsync(&hf, fl, sizeof(Flags), hipMemcpyDeviceToHost, st));
    TD_OK(hipStreamSynchronize(st));
    if (hf.bad || hf.maxEdgeFaces > kMaxEdgeFaces || hf.maxPointPoints > 255) return 1;      // (the host build handles it, or words the error)
    { const size_t n = (size_t)nEC; bg.th.emplace_back([&t, n] { resizeHuge(t.edgeCells.val, n); }); }
    int* ecCell = D.get<int>((size_t)nEC, why);
    uint8_t *ecF0 = D.get<uint8_t>((size_t)nEC, why), *ecF1 = D.get<uint8_t>((size_t)nEC, why);
    int *ringFace = D.get<int>((size_t)nnz, why), *ringCell = D.get<int>((size_t)nEC, why);
    uint8_t* ringOk = D.get<uint8_t>((size_t)nE, why);
    if (!ecCell || !ecF0 || !ecF1 || !ringFace || !ringCell || !ringOk) return 2;
    hipLaunchKernelGGL(k_td_edgeCells<false>, dim3(gridOf(nE)), dim3(kTB), 0, st, nE, nIF, efOff, efFace, dOwn, dNei, (int*)nullptr, ecOff, ecCell, ecF0, ecF1, fl);
    hipLaunchKernelGGL(k_td_rings, dim3(gridOf(nE)), dim3(kTB), 0, st, nE, efOff, efFace, ecOff, ecCell, ecF0, ecF1, ringFace, ringCell, ringOk);
    TD_OK(hipMemcpyAsync(&hf, fl, sizeof(Flags), hipMemcpyDeviceToHost, st));
    TD_OK(hipStreamSynchronize(st));
    if (hf.bad) return 1;
    lap("edgeCells + rings");

    // ---- the host's copy (tile tables, halo tables, layers / boundary set-up and the debug getters read it) ----
    // In three stages, each followed by the hook that lets a tile-table build start on what has arrived (as Topology::build's
    // hooks do); within a stage the arrays are sized by one thread each (first touch of 1.6 GB of fresh pages) and copied in
    // 32 MB pieces by four threads with a stream each (one pageable copy at a time ran at ~1.1 GB/s: 1.4 s of the 1.65 s this
    // function took for 10 M cells).
    t.nPoints = nP; t.nCells = nC; t.nFaces = nF; t.nInternalFaces = nIF; t.nEdges = nE;
    t.maxFaceSize = hf.maxFace; t.maxEdgeFaces = hf.maxEdgeFaces; t.maxPointCells = hf.maxPointCells; t.maxPointPoints = hf.maxPointPoints;
    std::vector<DownJob> jobs;
    auto want = [&](auto& vec, const auto* src, size_t n) { wantDown(jobs, vec, src, n); };
    bool copyFailed = false;
    auto flush = [&]() { if (!runDown(jobs, device)) copyFailed = true; };
    if (keep) {      // what the kernels read as it is stays on the device, owned by the caller -- from here on (the hooks below start
                     // tile-table builds that read these arrays); a later failure leaves the caller to free them
        auto give = [&](DeviceTopologyArrays::Arr& a, void* p, size_t bytes) { a.p = p; a.bytes = std::max<size_t>(bytes, 1); D.release(p); };
        give(keep->faceOff, dFo, ((size_t)nF + 1) * 4); give(keep->facePts, dFp, (size_t)nnz * 4);
        give(keep->cfOff, cfOff, ((size_t)nC + 1) * 4); give(keep->cfVal, cfVal, (size_t)nCF * 4);
        give(keep->pcOff, pcOff, ((size_t)nP + 1) * 4); give(keep->pcVal, pcVal, (size_t)nPC * 4);
        give(keep->ppOff, ppOff, ((size_t)nP + 1) * 4); give(keep->ppPt, ppPt, 2 * (size_t)nE * 4); give(keep->peEdge, peEdge, 2 * (size_t)nE * 4);
        give(keep->pfOff, pfOff, ((size_t)nP + 1) * 4); give(keep->pfFace, pfFace, (size_t)nnz * 4); give(keep->pfPrev, pfPrev, (size_t)nnz * 4); give(keep->pfNext, pfNext, (size_t)nnz * 4);
        give(keep->pfPrevSlot, prevSlot, (size_t)nnz); give(keep->pfNextSlot, nextSlot, (size_t)nnz);
        give(keep->ringFace, ringFace, (size_t)nnz * 4); give(keep->ringCell, ringCell, (size_t)nEC * 4); give(keep->edgeRingOk, ringOk, (size_t)nE);
        give(keep->edges, edges, 2 * (size_t)nE * 4); give(keep->efOff, efOff, ((size_t)nE + 1) * 4); give(keep->efFace, efFace, (size_t)nnz * 4);
        give(keep->ecOff, ecOff, ((size_t)nE + 1) * 4); give(keep->ecCell, ecCell, (size_t)nEC * 4); give(keep->ecF0, ecF0, (size_t)nEC); give(keep->ecF1, ecF1, (size_t)nEC);
        give(keep->owner, dOwn, (size_t)nF * 4); give(keep->neighbour, dNei, (size_t)std::max(nIF, 1) * 4);
        keep->valid = true;
    }
    // (t.facePoints / owner / neighbour: copied by the background threads; the first three of them are waited for here, the
    // sizing of the later stages' lists goes on beside this stage's copy)
    for (size_t i = 0; i < 3 && i < bg.th.size(); ++i) if (bg.th[i].joinable()) bg.th[i].join();
    want(t.cellFacesGeom.off, cfOff, (size_t)nC + 1); want(t.cellFacesGeom.val, cfVal, (size_t)nCF);
    flush();
    if (copyFailed) { why = "device -> host copy of the addressing failed"; return 2; }
    lap("download: cell lists");
    if (afterCells) afterCells();
    // (the order of the stages: what the three tile-boundary passes read first -- the longest of them, the edge pass, early --
    // then the lists only the halo / layer set-up and the getters read)
    bg.join();
    want(t.edges, edges, 2 * (size_t)nE);
    want(t.edgeFaces.off, efOff, (size_t)nE + 1); want(t.edgeFaces.val, efFace, (size_t)nnz);
    want(t.edgeCells.off, ecOff, (size_t)nE + 1); want(t.edgeCells.val, ecCell, (size_t)nEC);
    flush();
    if (copyFailed) { why = "device -> host copy of the addressing failed"; return 2; }
    lap("download: edge lists");
    if (afterEdges) afterEdges();
    want(t.pointCells.off, pcOff, (size_t)nP + 1); want(t.pointCells.val, pcVal, (size_t)nPC);
    want(t.pointEdges.off, ppOff, (size_t)nP + 1); want(t.pointPoints, ppPt, 2 * (size_t)nE);
    flush();
    if (copyFailed) { why = "device -> host copy of the addressing failed"; return 2; }
    lap("download: point lists");
    if (afterPoints) afterPoints();
    if (!(deferUnread && keep)) {      // (else fetched when someone asks: downloadDeferredLists)
        want(t.pointFaces.off, pfOff, (size_t)nP + 1); want(t.pointFaces.val, pfFace, (size_t)nnz);
        want(t.pfPrev, pfPrev, (size_t)nnz); want(t.pfNext, pfNext, (size_t)nnz);
        want(t.pointEdges.val, peEdge, 2 * (size_t)nE);
        want(t.pfPrevSlot, prevSlot, (size_t)nnz); want(t.pfNextSlot, nextSlot, (size_t)nnz);
        want(t.ecFace0, ecF0, (size_t)nEC); want(t.ecFace1, ecF1, (size_t)nEC);
        want(t.ringFace, ringFace, (size_t)nnz); want(t.ringCell, ringCell, (size_t)nEC); want(t.edgeRingOk, ringOk, (size_t)nE);
        flush();
        if (copyFailed) { why = "device -> host copy of the addressing failed"; return 2; }
    }
    lap("download: the rest");
    return 0;
}

int downloadDeferredLists(Topology& t, const DeviceTopologyArrays& td, int device, std::string& why, int groups) {
    if (!td.valid) return 0;
    std::vector<DownJob> jobs;
    auto fetch = [&](auto& vec, const DeviceTopologyArrays::Arr& a, size_t n) {
        typedef std::remove_reference_t<decltype(vec[0])> T;
        if (vec.size() != n) wantDown(jobs, vec, (const T*)a.p, n);
    };
    const size_t nnz = t.facePoints.val.size(), nEC = t.edgeCells.val.size(), nP = (size_t)t.nPoints, nE = (size_t)t.nEdges;
    if (groups & 1) {
        fetch(t.pointFaces.off, td.pfOff, nP + 1); fetch(t.pointFaces.val, td.pfFace, nnz); fetch(t.pfPrev, td.pfPrev, nnz); fetch(t.pfNext, td.pfNext, nnz);
        fetch(t.pointEdges.val, td.peEdge, 2 * nE);
    }
    if (groups & 2) {
        fetch(t.pfPrevSlot, td.pfPrevSlot, nnz); fetch(t.pfNextSlot, td.pfNextSlot, nnz);
        fetch(t.ecFace0, td.ecF0, nEC); fetch(t.ecFace1, td.ecF1, nEC);
        fetch(t.ringFace, td.ringFace, nnz); fetch(t.ringCell, td.ringCell, nEC); fetch(t.edgeRingOk, td.edgeRingOk, nE);
    }
    if (!jobs.empty() && !runDown(jobs, device)) { why = "device -> host copy of the addressing failed"; return 2; }
    return 0;
}

}  // namespace smgpu
