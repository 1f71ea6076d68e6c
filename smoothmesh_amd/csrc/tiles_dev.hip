// tiles_dev.hip -- the per-tile TABLES of tiles.cpp built on the device (round 5).  The tile boundaries stay the host's greedy pass
// (tiles.cpp, on segments of the Z-curve); given them, a tile's tables are a pure function of the addressing: the sorted unique ids
// of the records it stages, every list entry's position in those (binary search), the sliced-ELL rows.  One workgroup per tile:
// candidates into LDS, bitonic sort, duplicates out, searches -- twice, once to COUNT (the tables' offsets are prefix sums over the
// tiles) and once to FILL.  The host spent 1.0 / 1.2 / 0.9 s per table set for 10 M cells on 32 threads (1.5 / 2.2 / 0.9 s side by
// side); here a set is a few milliseconds, and the kernels read the arrays where they were built.  Same bytes as the host build
// (tests/test_gpu_topology.py compares checksums of every table of both builds); a tile the buffers below cannot hold hands the whole
// set back to the host build.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "tiles.hpp"

namespace smgpu {

namespace {

constexpr int kT = 256;                 // threads per workgroup = threads per tile of the kernels that use the tables
constexpr int kBuf = 4096;              // candidates per list a tile may have before duplicates are removed
constexpr int kPadKey = 0x7fffffff;
typedef unsigned long long u64;

#define TL_OK(expr)                                                                                             \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) { why = std::string(#expr) + ": " + hipGetErrorString(e__); return 2; }          \
    } while (0)

// exclusive prefix sum over the workgroup; total = sum of all (all threads call)
__device__ __forceinline__ int blockExclusive(int v, int* sh /* [kT / 64 + 1] */, int& total) {
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += t; }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += sh[i];
    total = 0;
    for (int i = 0; i < kT / 64; ++i) total += sh[i];
    return base + incl - v;
}
__device__ __forceinline__ int blockMax(int v, int* sh) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int m = sh[0];
    for (int i = 1; i < kT / 64; ++i) m = max(m, sh[i]);
    return m;
}
// buf[0 .. n) ascending without duplicates; returns the new length.  buf has room for the next power of two >= n (<= kBuf).
__device__ __forceinline__ int sortUnique(int* buf, int n, int* sh) {
    int N = 64;
    while (N < n) N <<= 1;
    for (int i = n + (int)threadIdx.x; i < N; i += kT) buf[i] = kPadKey;
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += kT) {
                const int l = i ^ j;
                if (l > i) {
                    const int a = buf[i], b = buf[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { buf[i] = b; buf[l] = a; }
                }
            }
            __syncthreads();
        }
    // duplicates out: every thread takes a contiguous chunk, counts its heads, scans, and writes them into the front -- in two
    // steps through registers (a chunk's heads never land behind the chunk's own start, but may land in an earlier chunk's range)
    const int per = (n + kT - 1) / kT, lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int heads = 0;
    for (int i = lo; i < hi; ++i) heads += (i == 0 || buf[i] != buf[i - 1]) ? 1 : 0;
    int total;
    const int at = blockExclusive(heads, sh, total);
    int keep[kBuf / kT];      // (per <= kBuf / kT)
    int c = 0;
    for (int i = lo; i < hi; ++i) if (i == 0 || buf[i] != buf[i - 1]) keep[c++] = buf[i];
    __syncthreads();
    for (int i = 0; i < c; ++i) buf[at + i] = keep[i];
    __syncthreads();
    return total;
}
__device__ __forceinline__ int lowerBound(const int* v, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int roundUp4d(int v) { return (v + 3) & ~3; }

// ---- geometry tiles ------------------------------------------------------------------------------------------------------------
struct GeomIn {
    const int* order; const int* cellBeg; int nTiles;
    const int* cfOff; const int* cfVal; const int* faceOff; const int* facePts; const int* owner; const int* cellTile;
};
struct GeomSizes { int* nPts; int* nFcs; int* fw; int* cw; int* flags; int* bad; };
struct GeomOut {
    const long long* tpOff; const long long* tfOff; const long long* fvBase; const long long* cfBase;
    int* tpIds; int* tfIds; uint16_t* faceVerts; uint16_t* cellFaces; int* meta;
};
__global__ void __launch_bounds__(kT) k_tl_cellTile(GeomIn in, int* cellTile) {
    const int ti = blockIdx.x, cb = in.cellBeg[ti], n = in.cellBeg[ti + 1] - cb;
    for (int i = threadIdx.x; i < n; i += kT) cellTile[in.order[cb + i]] = ti;
}
template <bool FILL>
__global__ void __launch_bounds__(kT) k_tl_geom(GeomIn in, GeomSizes sz, GeomOut out) {
    __shared__ int sFaces[kBuf], sPts[kBuf], sh[kT / 64 + 1];
    const int ti = blockIdx.x, tid = threadIdx.x;
    const int cb = in.cellBeg[ti], nC = in.cellBeg[ti + 1] - cb;
    // the tile's faces
    const int c = (tid < nC) ? in.order[cb + tid] : -1;
    const int fb = (c >= 0) ? in.cfOff[c] : 0, nfc = (c >= 0) ? in.cfOff[c + 1] - fb : 0;
    int rawF;
    const int atF = blockExclusive(nfc, sh, rawF);
    if (rawF > kBuf || nC > kT) { if (tid == 0) *sz.bad = 1; return; }      // (workgroup-uniform)
    for (int k = 0; k < nfc; ++k) sFaces[atF + k] = in.cfVal[fb + k] & 0x7fffffff;
    __syncthreads();
    const int nF = sortUnique(sFaces, rawF, sh);
    // ... and their points
    int rawP = 0;
    int myCount = 0;
    for (int i = tid; i < nF; i += kT) myCount += in.faceOff[sFaces[i] + 1] - in.faceOff[sFaces[i]];
    int atP = blockExclusive(myCount, sh, rawP);
    if (rawP > kBuf) { if (tid == 0) *sz.bad = 1; return; }
    int fwMax = 0;
    bool quads = true;
    for (int i = tid; i < nF; i += kT) {
        const int f = sFaces[i], b = in.faceOff[f], nv = in.faceOff[f + 1] - b;
        fwMax = max(fwMax, nv);
        quads = quads && nv == 4;
        for (int k = 0; k < nv; ++k) sPts[atP++] = in.facePts[b + k];
    }
    __syncthreads();
    const int nP = sortUnique(sPts, rawP, sh);
    const int fw = roundUp4d(blockMax(fwMax, sh)), cw = roundUp4d(blockMax(nfc, sh));
    const int notQuads = blockMax(quads ? 0 : 1, sh), notHex = blockMax((c >= 0 && nfc != 6) ? 1 : 0, sh);
    if (nP > 32767 || nF > 32767 || cw > 252 || fw > 252) { if (tid == 0) *sz.bad = 1; return; }
    if (!FILL) {
        if (tid == 0) { sz.nPts[ti] = nP; sz.nFcs[ti] = nF; sz.fw[ti] = fw; sz.cw[ti] = cw; sz.flags[ti] = (notQuads ? 0 : 1) | (notHex ? 0 : 2); }
        return;
    }
    const long long tp = out.tpOff[ti], tf = out.tfOff[ti], fv = out.fvBase[ti], cfb = out.cfBase[ti];
    for (int i = tid; i < nP; i += kT) out.tpIds[tp + i] = sPts[i];
    for (int i = tid; i < nF; i += kT) {
        const int f = sFaces[i];
        out.tfIds[tf + i] = (in.cellTile[in.owner[f]] == ti) ? (int)(0x80000000u | (unsigned)f) : f;
        const int b = in.faceOff[f], nv = in.faceOff[f + 1] - b;
        uint16_t* row = out.faceVerts + fv + (long long)i * fw;
        for (int j = 0; j < fw; ++j) row[j] = (j < nv) ? (uint16_t)lowerBound(sPts, nP, in.facePts[b + j]) : (uint16_t)0xFFFF;
    }
    // cell faces: sliced ELL, entry (j, t) at ((j / 4) * T + t) * 4 + j % 4; the rows were filled with pads by a memset
    if (c >= 0)
        for (int j = 0; j < nfc; ++j) {
            const int v = in.cfVal[fb + j];
            out.cellFaces[cfb + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)(lowerBound(sFaces, nF, v & 0x7fffffff) | (v < 0 ? 0x8000 : 0));
        }
    if (tid == 0) {
        int* r = out.meta + 12ll * ti;
        r[0] = (int)tp; r[1] = nP; r[2] = (int)tf; r[3] = nF; r[4] = (int)fv; r[5] = fw; r[6] = cb; r[7] = nC; r[8] = (int)cfb; r[9] = cw;
        r[10] = (notQuads ? 0 : 1) | (notHex ? 0 : 2); r[11] = 0;
    }
}
// per tile: the four running sums the tables' offsets are
__global__ void __launch_bounds__(kT) k_tl_geomTerms(int nTiles, GeomSizes sz, long long* a, long long* b, long long* c, long long* d) {
    const int ti = blockIdx.x * kT + threadIdx.x;
    if (ti >= nTiles) return;
    a[ti] = sz.nPts[ti]; b[ti] = sz.nFcs[ti]; c[ti] = (long long)sz.nFcs[ti] * sz.fw[ti]; d[ti] = (long long)sz.cw[ti] * kT;
}


// ---- edge tiles (face-angle filter) -------------------------------------------------------------------------------------------------
struct EdgeIn {
    const int* order; const int* edgeBeg; int nTiles;
    const int* edges; const int* efOff; const int* efFace; const int* ecOff; const int* ecCell; const int* ringFace; const int* ringCell; const uint8_t* ringOk;
};
struct EdgeSizes { int* nPts; int* nFcs; int* nCls; int* wf; int* wc; int* bad; };
struct EdgeOut {
    const long long* tpOff; const long long* tfOff; const long long* tcOff; const long long* efBase; const long long* ecBase;
    int* tpIds; int* tfIds; int* tcIds; uint16_t* epLoc; uint16_t* efEll; uint16_t* ecEll; int* meta;
};
template <bool FILL>
__global__ void __launch_bounds__(kT) k_tl_edge(EdgeIn in, EdgeSizes sz, EdgeOut out) {
    __shared__ int sFcs[kBuf], sCls[kBuf], sPts[2 * kT], sh[kT / 64 + 1];
    const int ti = blockIdx.x, tid = threadIdx.x;
    const int eb = in.edgeBeg[ti], nEd = in.edgeBeg[ti + 1] - eb;
    if (nEd > kT) { if (tid == 0) *sz.bad = 1; return; }
    const int e = (tid < nEd) ? in.order[eb + tid] : -1;
    const int fb = (e >= 0) ? in.efOff[e] : 0, nf = (e >= 0) ? in.efOff[e + 1] - fb : 0;
    const int cb = (e >= 0) ? in.ecOff[e] : 0, nc = (e >= 0) ? in.ecOff[e + 1] - cb : 0;
    int rawF, rawC;
    const int atF = blockExclusive(nf, sh, rawF);
    const int atC = blockExclusive(nc, sh, rawC);
    if (rawF > kBuf || rawC > kBuf) { if (tid == 0) *sz.bad = 1; return; }
    for (int k = 0; k < nf; ++k) sFcs[atF + k] = in.efFace[fb + k];
    for (int k = 0; k < nc; ++k) sCls[atC + k] = in.ecCell[cb + k];
    if (e >= 0) { sPts[2 * tid] = in.edges[2 * e]; sPts[2 * tid + 1] = in.edges[2 * e + 1]; }
    __syncthreads();
    const int nF = sortUnique(sFcs, rawF, sh);
    const int nC = sortUnique(sCls, rawC, sh);
    const int nP = sortUnique(sPts, 2 * nEd, sh);
    const int wf = roundUp4d(blockMax(nf, sh)), wc = roundUp4d(blockMax(nc, sh));
    if (nP > 32766 || nF > 32766 || nC > 32766 || wf > 252 || wc > 252) { if (tid == 0) *sz.bad = 1; return; }
    if (!FILL) {
        if (tid == 0) { sz.nPts[ti] = nP; sz.nFcs[ti] = nF; sz.nCls[ti] = nC; sz.wf[ti] = wf; sz.wc[ti] = wc; }
        return;
    }
    const long long tp = out.tpOff[ti], tf = out.tfOff[ti], tc = out.tcOff[ti], efb = out.efBase[ti], ecb = out.ecBase[ti];
    for (int i = tid; i < nP; i += kT) out.tpIds[tp + i] = sPts[i];
    for (int i = tid; i < nF; i += kT) out.tfIds[tf + i] = sFcs[i];
    for (int i = tid; i < nC; i += kT) out.tcIds[tc + i] = sCls[i];
    if (e >= 0) {
        out.epLoc[2 * (long long)(eb + tid)] = (uint16_t)lowerBound(sPts, nP, in.edges[2 * e]);
        out.epLoc[2 * (long long)(eb + tid) + 1] = (uint16_t)lowerBound(sPts, nP, in.edges[2 * e + 1]);
        if (in.ringOk[e]) {      // (else all-pad rows: the filter flags the edge UNSURE)
            for (int j = 0; j < nf; ++j) out.efEll[efb + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)lowerBound(sFcs, nF, in.ringFace[fb + j]);
            for (int j = 0; j < nc; ++j) out.ecEll[ecb + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)lowerBound(sCls, nC, in.ringCell[cb + j]);
        }
    }
    if (tid == 0) {
        int* r = out.meta + 12ll * ti;
        r[0] = eb; r[1] = nEd; r[2] = (int)tp; r[3] = nP; r[4] = (int)tf; r[5] = nF; r[6] = (int)tc; r[7] = nC; r[8] = (int)efb; r[9] = wf; r[10] = (int)ecb; r[11] = wc;
    }
}
__global__ void __launch_bounds__(kT) k_tl_edgeTerms(int nTiles, EdgeSizes sz, long long* a, long long* b, long long* c, long long* d, long long* e) {
    const int ti = blockIdx.x * kT + threadIdx.x;
    if (ti >= nTiles) return;
    a[ti] = sz.nPts[ti]; b[ti] = sz.nFcs[ti]; c[ti] = sz.nCls[ti]; d[ti] = (long long)sz.wf[ti] * kT; e[ti] = (long long)sz.wc[ti] * kT;
}
// the face ids of the edge tiles as positions in the geometry tiles' face lists (where the geometry kernel stores the face vertex
// averages the filter reads): facePos[f] = position of f in its owner's tile
__global__ void __launch_bounds__(kT) k_tl_facePos(long long n, const int* __restrict__ geomTfIds, int* __restrict__ facePos) {
    const long long k = (long long)blockIdx.x * kT + threadIdx.x;
    if (k < n && geomTfIds[k] < 0) facePos[geomTfIds[k] & 0x7fffffff] = (int)k;
}
__global__ void __launch_bounds__(kT) k_tl_remap(long long n, int* __restrict__ ids, const int* __restrict__ facePos) {
    const long long k = (long long)blockIdx.x * kT + threadIdx.x;
    if (k < n) ids[k] = facePos[ids[k]];
}

// ---- smoothing tiles ----------------------------------------------------------------------------------------------------------------
// The face corners of every point in the order of Euler trails: tiles.cpp's chainCorners, statement for statement (the order is part
// of the tables' bytes), one thread per point on arrays in its private memory.  It depends on the point only, not on the tiling:
// one pass over the points, the tile kernel below maps the result to local indices.
constexpr int kCh = 64;                 // corners (= faces) per point the kernels handle; a mesh beyond goes back to the host build
constexpr int kChLds = 16;              // ... per point of the LDS form (hex meshes: 12, the castellated polyhedral meshes: <= 16 for nearly all)
// the working arrays of one point's walk, as the two kernels below hold them: arrays in the thread's private (scratch) memory, or
// columns of LDS arrays (element i of lane l at [i][l]: bank-conflict free)
template <int CH>
struct ChainPrivate {
    int verts_[2 * CH];
    uint8_t ea_[2 * CH], eb_[2 * CH], deg_[2 * CH], used_[2 * CH], stackV_[2 * CH + 1], compE_[2 * CH], compV_[2 * CH], adjEdge_[4 * CH];
    short stackE_[2 * CH + 1];
    unsigned short adjOff_[2 * CH + 1], next_[2 * CH];
    __device__ __forceinline__ int& verts(int i) { return verts_[i]; }
    __device__ __forceinline__ uint8_t& ea(int i) { return ea_[i]; }
    __device__ __forceinline__ uint8_t& eb(int i) { return eb_[i]; }
    __device__ __forceinline__ uint8_t& deg(int i) { return deg_[i]; }
    __device__ __forceinline__ uint8_t& used(int i) { return used_[i]; }
    __device__ __forceinline__ uint8_t& stackV(int i) { return stackV_[i]; }
    __device__ __forceinline__ uint8_t& compE(int i) { return compE_[i]; }
    __device__ __forceinline__ uint8_t& compV(int i) { return compV_[i]; }
    __device__ __forceinline__ uint8_t& adjEdge(int i) { return adjEdge_[i]; }
    __device__ __forceinline__ short& stackE(int i) { return stackE_[i]; }
    __device__ __forceinline__ unsigned short& adjOff(int i) { return adjOff_[i]; }
    __device__ __forceinline__ unsigned short& next(int i) { return next_[i]; }
};
template <int CH>
struct ChainLdsBlock {       // one wave's arrays (64 lanes)
    int verts[2 * CH][64];
    uint8_t ea[2 * CH][64], eb[2 * CH][64], deg[2 * CH][64], used[2 * CH][64], stackV[2 * CH + 1][64], compE[2 * CH][64], compV[2 * CH][64], adjEdge[4 * CH][64];
    short stackE[2 * CH + 1][64];
    unsigned short adjOff[2 * CH + 1][64], next[2 * CH][64];
};
template <int CH>
struct ChainLds {
    ChainLdsBlock<CH>* B; int l;
    __device__ __forceinline__ int& verts(int i) { return B->verts[i][l]; }
    __device__ __forceinline__ uint8_t& ea(int i) { return B->ea[i][l]; }
    __device__ __forceinline__ uint8_t& eb(int i) { return B->eb[i][l]; }
    __device__ __forceinline__ uint8_t& deg(int i) { return B->deg[i][l]; }
    __device__ __forceinline__ uint8_t& used(int i) { return B->used[i][l]; }
    __device__ __forceinline__ uint8_t& stackV(int i) { return B->stackV[i][l]; }
    __device__ __forceinline__ uint8_t& compE(int i) { return B->compE[i][l]; }
    __device__ __forceinline__ uint8_t& compV(int i) { return B->compV[i][l]; }
    __device__ __forceinline__ uint8_t& adjEdge(int i) { return B->adjEdge[i][l]; }
    __device__ __forceinline__ short& stackE(int i) { return B->stackE[i][l]; }
    __device__ __forceinline__ unsigned short& adjOff(int i) { return B->adjOff[i][l]; }
    __device__ __forceinline__ unsigned short& next(int i) { return B->next[i][l]; }
};
template <class W>
__device__ __forceinline__ int chainLowerBound(W& w, int n, int x) {      // (lowerBound on w.verts)
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (w.verts(mid) < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// the chain of point p's n >= 2 corners (rows b .. b + n of pfPrev / pfNext) into chPrev / chNext.  The trail's edges go straight
// to the output rows (there are never more than n of them); a walk that does not return all n corners leaves the input order.
template <class W>
__device__ __forceinline__ void chainOfPoint(W& w, int b, int n, const int* __restrict__ pfPrev, const int* __restrict__ pfNext, int* __restrict__ chPrev, int* __restrict__ chNext) {
    for (int k = 0; k < n; ++k) { w.verts(2 * k) = pfPrev[b + k]; w.verts(2 * k + 1) = pfNext[b + k]; }
    for (int i = 1; i < 2 * n; ++i) { const int x = w.verts(i); int j = i - 1; while (j >= 0 && w.verts(j) > x) { w.verts(j + 1) = w.verts(j); --j; } w.verts(j + 1) = x; }
    int nv = 0;
    for (int i = 0; i < 2 * n; ++i) if (i == 0 || w.verts(i) != w.verts(i - 1)) { const int x = w.verts(i); w.verts(nv++) = x; }
    for (int k = 0; k < n; ++k) { w.ea(k) = (uint8_t)chainLowerBound(w, nv, pfPrev[b + k]); w.eb(k) = (uint8_t)chainLowerBound(w, nv, pfNext[b + k]); }
    for (int v = 0; v < nv; ++v) w.deg(v) = 0;
    for (int e = 0; e < n; ++e) { ++w.deg(w.ea(e)); ++w.deg(w.eb(e)); }
    int ne = n, pending = -1;
    for (int v = 0; v < nv; ++v)
        if (w.deg(v) & 1) {
            if (pending < 0) pending = v;
            else { w.ea(ne) = (uint8_t)pending; w.eb(ne) = (uint8_t)v; ++ne; ++w.deg(pending); ++w.deg(v); pending = -1; }
        }
    for (int v = 0; v <= nv; ++v) w.adjOff(v) = 0;
    for (int e = 0; e < ne; ++e) { ++w.adjOff(w.ea(e) + 1); ++w.adjOff(w.eb(e) + 1); }
    for (int v = 0; v < nv; ++v) w.adjOff(v + 1) = (unsigned short)(w.adjOff(v + 1) + w.adjOff(v));
    for (int v = 0; v < nv; ++v) w.next(v) = w.adjOff(v);
    for (int e = 0; e < ne; ++e) { w.adjEdge(w.next(w.ea(e))++) = (uint8_t)e; w.adjEdge(w.next(w.eb(e))++) = (uint8_t)e; }
    for (int e = 0; e < ne; ++e) w.used(e) = 0;
    for (int v = 0; v < nv; ++v) w.next(v) = w.adjOff(v);
    int nOut = 0;
    for (int start = 0; start < nv; ++start) {
        if (w.next(start) >= w.adjOff(start + 1)) continue;
        int sp = 1, nComp = 0;
        w.stackV(0) = (uint8_t)start; w.stackE(0) = -1;
        while (sp > 0) {
            const int v = w.stackV(sp - 1);
            int e = -1;
            while (w.next(v) < w.adjOff(v + 1)) {
                const int cand = w.adjEdge(w.next(v)++);
                if (!w.used(cand)) { e = cand; break; }
            }
            if (e >= 0) {
                w.used(e) = 1;
                const int to = (w.ea(e) == v) ? w.eb(e) : w.ea(e);
                w.stackV(sp) = (uint8_t)to; w.stackE(sp) = (short)e; ++sp;
            } else {
                const int eIn = w.stackE(sp - 1);
                --sp;
                if (eIn >= 0) { w.compE(nComp) = (uint8_t)eIn; w.compV(nComp) = w.stackV(sp - 1); ++nComp; }
            }
        }
        for (int i = nComp; i-- > 0;) {
            const int e = w.compE(i), from = w.compV(i);
            if (e >= n) continue;
            const int to = (w.ea(e) == from) ? w.eb(e) : w.ea(e);
            if (nOut < n) { chPrev[b + nOut] = w.verts(from); chNext[b + nOut] = w.verts(to); }
            ++nOut;
        }
    }
    if (nOut != n) for (int k = 0; k < n; ++k) { chPrev[b + k] = pfPrev[b + k]; chNext[b + k] = pfNext[b + k]; }
}
// One thread per point, its arrays as LDS columns (a sequential graph walk per thread: on arrays in scratch memory -- 3 KB per
// lane -- the kernel took 180 ms for the 10 M points of the cavity mesh, every step a round trip beyond the L1).  Points with more
// than kChLds corners go to `big` (count in big[-1 .. ]: bigCount) for k_tl_chain_big.
__global__ void __launch_bounds__(64) k_tl_chain(int nPos, const int* __restrict__ ids /* the points to do, or NULL: all */, const int* __restrict__ pfOff,
                                                 const int* __restrict__ pfPrev, const int* __restrict__ pfNext, int* __restrict__ chPrev, int* __restrict__ chNext,
                                                 int* __restrict__ big, int* __restrict__ bigCount) {
    __shared__ ChainLdsBlock<kChLds> blk;
    const int i_ = blockIdx.x * 64 + threadIdx.x;
    if (i_ >= nPos) return;
    const int p = ids ? ids[i_] : i_;
    const int b = pfOff[p], n = pfOff[p + 1] - b;
    if (n > kChLds) { big[atomicAdd(bigCount, 1)] = p; return; }
    if (n < 2) { for (int k = 0; k < n; ++k) { chPrev[b + k] = pfPrev[b + k]; chNext[b + k] = pfNext[b + k]; } return; }
    ChainLds<kChLds> w{&blk, (int)threadIdx.x};
    chainOfPoint(w, b, n, pfPrev, pfNext, chPrev, chNext);
}
// the points k_tl_chain left (more than kChLds corners), on private arrays; beyond kCh corners the mesh goes back to the host build
__global__ void __launch_bounds__(64) k_tl_chain_big(const int* __restrict__ big, const int* __restrict__ bigCount, const int* __restrict__ pfOff,
                                                     const int* __restrict__ pfPrev, const int* __restrict__ pfNext, int* __restrict__ chPrev, int* __restrict__ chNext, int* bad) {
    const int nBig = *bigCount;
    for (int i_ = blockIdx.x * 64 + threadIdx.x; i_ < nBig; i_ += gridDim.x * 64) {
        const int p = big[i_];
        const int b = pfOff[p], n = pfOff[p + 1] - b;
        if (n > kCh) { *bad = 1; continue; }
        ChainPrivate<kCh> w;
        chainOfPoint(w, b, n, pfPrev, pfNext, chPrev, chNext);
    }
}
// both launches on stream st; big = nPos + 1 ints of scratch (the count in big[nPos])
static void launchChains(hipStream_t st, int nPos, const int* ids, const int* pfOff, const int* pfPrev, const int* pfNext, int* chPrev, int* chNext, int* big, int* bad) {
    (void)hipMemsetAsync(big + nPos, 0, 4, st);
    hipLaunchKernelGGL(k_tl_chain, dim3((nPos + 63) / 64), dim3(64), 0, st, nPos, ids, pfOff, pfPrev, pfNext, chPrev, chNext, big, big + nPos);
    hipLaunchKernelGGL(k_tl_chain_big, dim3(std::max(1, std::min((nPos + 63) / 64, 2048))), dim3(64), 0, st, (const int*)big, (const int*)(big + nPos), pfOff, pfPrev, pfNext, chPrev, chNext, bad);
}

struct SmoothIn {
    const int* order; const int* ptBeg; int nTiles; int pairs;
    const int* pcOff; const int* pcVal; const int* ppOff; const int* ppPt; const int* pfOff; const int* chPrev; const int* chNext; const uint8_t* isInternal;
};
struct SmoothSizes { int* nCl; int* nPt; int* wc; int* wn; int* wf; int* bad; };
struct SmoothOut {
    const long long* tcOff; const long long* tnOff; const long long* pcBase; const long long* ppBase; const long long* pfBase;
    int* tcIds; int* tnIds; uint16_t* selfLoc; uint16_t* pcEll; uint16_t* ppEll; uint16_t* pairEll; uint16_t* pfEll; int* meta;
};
template <bool FILL>
__global__ void __launch_bounds__(kT) k_tl_smooth(SmoothIn in, SmoothSizes sz, SmoothOut out) {
    __shared__ int sCells[kBuf], sPts[kBuf], sh[kT / 64 + 1];
    const int ti = blockIdx.x, tid = threadIdx.x;
    const int pb = in.ptBeg[ti], nPos = in.ptBeg[ti + 1] - pb;
    if (nPos > kT) { if (tid == 0) *sz.bad = 1; return; }
    const int p = (tid < nPos) ? in.order[pb + tid] : -1;
    const int cb = (p >= 0) ? in.pcOff[p] : 0, nc = (p >= 0) ? in.pcOff[p + 1] - cb : 0;
    const int nb = (p >= 0) ? in.ppOff[p] : 0, nn = (p >= 0) ? in.ppOff[p + 1] - nb : 0;
    const int fb = (p >= 0) ? in.pfOff[p] : 0, nf = (p >= 0) ? in.pfOff[p + 1] - fb : 0;
    int rawC, rawN;
    const int atC = blockExclusive(nc, sh, rawC);
    const int atN = blockExclusive((p >= 0) ? nn + 1 : 0, sh, rawN);
    if (rawC > kBuf || rawN > kBuf) { if (tid == 0) *sz.bad = 1; return; }
    for (int k = 0; k < nc; ++k) sCells[atC + k] = in.pcVal[cb + k];
    if (p >= 0) { sPts[atN] = p; for (int k = 0; k < nn; ++k) sPts[atN + 1 + k] = in.ppPt[nb + k]; }
    __syncthreads();
    const int nC = sortUnique(sCells, rawC, sh);
    const int nN = sortUnique(sPts, rawN, sh);
    const int wc = roundUp4d(blockMax(nc, sh)), wn = roundUp4d(blockMax(nn, sh)), wf = roundUp4d(blockMax(2 * nf, sh));
    if (nC > 32766 || nN > 32766 || wc > 252 || wn > 252 || wf > 252) { if (tid == 0) *sz.bad = 1; return; }
    if (!FILL) {
        if (tid == 0) { sz.nCl[ti] = nC; sz.nPt[ti] = nN; sz.wc[ti] = wc; sz.wn[ti] = wn; sz.wf[ti] = wf; }
        return;
    }
    const long long tc = out.tcOff[ti], tn = out.tnOff[ti], cbase = out.pcBase[ti], nbase = out.ppBase[ti], fbase = out.pfBase[ti];
    for (int i = tid; i < nC; i += kT) out.tcIds[tc + i] = sCells[i];
    for (int i = tid; i < nN; i += kT) out.tnIds[tn + i] = sPts[i];
    if (p >= 0) {
        out.selfLoc[pb + tid] = (uint16_t)lowerBound(sPts, nN, p);
        for (int c = 0; c < nf; ++c) {
            const int j = 2 * c;
            out.pfEll[fbase + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)lowerBound(sPts, nN, in.chPrev[fb + c]);
            out.pfEll[fbase + ((long long)((j + 1) / 4) * kT + tid) * 4 + ((j + 1) % 4)] = (uint16_t)lowerBound(sPts, nN, in.chNext[fb + c]);
        }
        for (int j = 0; j < nc; ++j) out.pcEll[cbase + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)lowerBound(sCells, nC, in.pcVal[cb + j]);
        for (int j = 0; j < nn; ++j) {
            const int q = in.ppPt[nb + j];
            out.ppEll[nbase + ((long long)(j / 4) * kT + tid) * 4 + (j % 4)] = (uint16_t)(lowerBound(sPts, nN, q) | (in.isInternal[q] ? 0x8000 : 0));
        }
        if (in.pairs) {      // neighbours i, j of p share a cell  <=>  pointCells(q_i) and pointCells(q_j) intersect (nn <= 16 here)
            unsigned masks[16];
            for (int i = 0; i < 16; ++i) masks[i] = 0;
            for (int i = 0; i < nn; ++i) {
                const int qi = in.ppPt[nb + i];
                const int a0 = in.pcOff[qi], ae = in.pcOff[qi + 1];
                for (int j = i + 1; j < nn; ++j) {
                    const int qj = in.ppPt[nb + j];
                    int a = a0, c = in.pcOff[qj];
                    const int ce = in.pcOff[qj + 1];
                    bool hit = false;
                    while (a < ae && c < ce) {
                        const int va = in.pcVal[a], vc = in.pcVal[c];
                        if (va == vc) { hit = true; break; }
                        if (va < vc) ++a; else ++c;
                    }
                    if (hit) { masks[i] |= 1u << j; masks[j] |= 1u << i; }
                }
            }
            for (int i = 0; i < nn; ++i) out.pairEll[nbase + ((long long)(i / 4) * kT + tid) * 4 + (i % 4)] = (uint16_t)masks[i];
        }
    }
    if (tid == 0) {
        int* r = out.meta + 12ll * ti;
        r[0] = pb; r[1] = nPos; r[2] = (int)tc; r[3] = nC; r[4] = (int)tn; r[5] = nN; r[6] = (int)cbase; r[7] = wc; r[8] = (int)nbase; r[9] = wn; r[10] = (int)fbase; r[11] = wf;
    }
}
__global__ void __launch_bounds__(kT) k_tl_smoothTerms(int nTiles, SmoothSizes sz, long long* a, long long* b, long long* c, long long* d, long long* e) {
    const int ti = blockIdx.x * kT + threadIdx.x;
    if (ti >= nTiles) return;
    a[ti] = sz.nCl[ti]; b[ti] = sz.nPt[ti]; c[ti] = (long long)sz.wc[ti] * kT; d[ti] = (long long)sz.wn[ti] * kT; e[ti] = (long long)sz.wf[ti] * kT;
}

// ---- the cells along the Z-curve (GeomTiles' order) ------------------------------------------------------------------------------
// tiles.cpp's mortonOrder on the cells' vertex averages, operation for operation: the average over the faces' points in
// cellFacesGeom order, one isotropic scale from the bounding box, 21 bits per axis interleaved, ties in id order (a stable sort of
// ids that start ascending).  0.28 s of the geometry tiles' chain on the host for 10 M cells, ~10 ms here.
__global__ void __launch_bounds__(kT) k_tl_cellAvg(int nCells, const int* __restrict__ cfOff, const int* __restrict__ cfVal, const int* __restrict__ faceOff,
                                                   const int* __restrict__ facePts, const double* __restrict__ pts, double* __restrict__ cc, double* __restrict__ part) {
    __shared__ double sh[6][kT / 64];
    const int c = blockIdx.x * kT + threadIdx.x;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    if (c < nCells) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        int n = 0;
        for (int k = cfOff[c]; k < cfOff[c + 1]; ++k) {
            const int f = cfVal[k] & 0x7fffffff;
            for (int j = faceOff[f]; j < faceOff[f + 1]; ++j, ++n) {
                const double* q = pts + 3 * (size_t)facePts[j];
                s0 += q[0]; s1 += q[1]; s2 += q[2];
            }
        }
        const double v[3] = {n ? s0 / n : 0.0, n ? s1 / n : 0.0, n ? s2 / n : 0.0};
        for (int a = 0; a < 3; ++a) { cc[3 * (size_t)c + a] = v[a]; lo[a] = v[a]; hi[a] = v[a]; }
    }
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o, 64)); }
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { sh[a][threadIdx.x >> 6] = lo[a]; sh[3 + a][threadIdx.x >> 6] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double m = sh[threadIdx.x][0];
        for (int i = 1; i < kT / 64; ++i) m = threadIdx.x < 3 ? fmin(m, sh[threadIdx.x][i]) : fmax(m, sh[threadIdx.x][i]);
        part[6 * (size_t)blockIdx.x + threadIdx.x] = m;
    }
}
__device__ __forceinline__ u64 spread21d(u64 v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__global__ void __launch_bounds__(kT) k_tl_mortonKeys(int n, const double* __restrict__ xyz, double lo0, double lo1, double lo2, double scale, u64* __restrict__ keys, int* __restrict__ ids) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const double lo[3] = {lo0, lo1, lo2};
    u64 k = 0;
    for (int a = 0; a < 3; ++a) k |= spread21d((u64)((xyz[3 * (size_t)i + a] - lo[a]) * scale)) << a;
    keys[i] = k;
    ids[i] = i;
}

struct DevBuf {
    std::vector<void*> all;
    ~DevBuf() { for (void* p : all) (void)hipFree(p); }
    template <class T> T* get(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        all.push_back(p);
        return (T*)p;
    }
    void release(void* p) { auto it = std::find(all.begin(), all.end(), p); if (it != all.end()) all.erase(it); }
};

}  // namespace

// gt.order / gt.cellBeg / gt.nTiles / gt.threads stand (host: GeomTiles::buildBoundaries).  0: the tables are in `out` (device) and
// gt holds what the host reads (offsets, widths, flags, tfIds, maxima); 1: not handled here (host tables); 2: a HIP error.
int buildGeomTablesOnDevice(GeomTiles& gt, const DeviceTopologyArrays& td, int32_t nCells, int device, GeomTilesDev& out, std::string& why) {
    if (!td.valid || !td.owner.p || gt.threads != kT || gt.nTiles <= 0) return 1;
    TL_OK(hipSetDevice(device));
    hipStream_t st = nullptr;
    TL_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};
    const int nT = gt.nTiles;
    DevBuf D;
    int *dOrder = D.get<int>((size_t)nCells), *dBeg = D.get<int>((size_t)nT + 1), *cellTile = D.get<int>((size_t)nCells);
    int *nPts = D.get<int>((size_t)nT), *nFcs = D.get<int>((size_t)nT), *fw = D.get<int>((size_t)nT), *cw = D.get<int>((size_t)nT), *flags = D.get<int>((size_t)nT), *bad = D.get<int>(1);
    long long* terms = D.get<long long>(8 * ((size_t)nT + 1));
    if (!dOrder || !dBeg || !cellTile || !nPts || !nFcs || !fw || !cw || !flags || !bad || !terms) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemcpyAsync(dOrder, gt.order.data(), (size_t)nCells * 4, hipMemcpyHostToDevice, st));
    TL_OK(hipMemcpyAsync(dBeg, gt.cellBeg.data(), ((size_t)nT + 1) * 4, hipMemcpyHostToDevice, st));
    TL_OK(hipMemsetAsync(bad, 0, 4, st));
    GeomIn in{dOrder, dBeg, nT, (const int*)td.cfOff.p, (const int*)td.cfVal.p, (const int*)td.faceOff.p, (const int*)td.facePts.p, (const int*)td.owner.p, cellTile};
    GeomSizes sz{nPts, nFcs, fw, cw, flags, bad};
    hipLaunchKernelGGL(k_tl_cellTile, dim3(nT), dim3(kT), 0, st, in, cellTile);
    hipLaunchKernelGGL(k_tl_geom<false>, dim3(nT), dim3(kT), 0, st, in, sz, GeomOut{});
    // offsets: exclusive scans of the four per-tile terms (64-bit: the totals are checked against int32 addressing below)
    long long *tA = terms, *tB = tA + nT + 1, *tC = tB + nT + 1, *tD = tC + nT + 1, *oA = tD + nT + 1, *oB = oA + nT + 1, *oC = oB + nT + 1, *oD = oC + nT + 1;
    TL_OK(hipMemsetAsync(terms, 0, 8 * ((size_t)nT + 1) * 8, st));
    hipLaunchKernelGGL(k_tl_geomTerms, dim3((nT + kT - 1) / kT), dim3(kT), 0, st, nT, sz, tA, tB, tC, tD);
    size_t tempBytes = 0;
    (void)rocprim::exclusive_scan(nullptr, tempBytes, tA, oA, 0ll, (size_t)nT + 1, rocprim::plus<long long>(), st);
    void* temp = D.get<char>(tempBytes + 256);
    if (!temp) { why = "device allocation failed"; return 2; }
    for (int q = 0; q < 4; ++q) {
        size_t b = tempBytes;
        TL_OK(rocprim::exclusive_scan(temp, b, tA + (size_t)q * (nT + 1), oA + (size_t)q * (nT + 1), 0ll, (size_t)nT + 1, rocprim::plus<long long>(), st));
    }
    std::vector<long long> offs(4 * ((size_t)nT + 1));
    int hbad = 0;
    TL_OK(hipMemcpyAsync(offs.data(), oA, offs.size() * 8, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    if (hbad) return 1;
    const long long nTp = offs[(size_t)nT], nTf = offs[(size_t)(nT + 1) + nT], nFv = offs[2 * (size_t)(nT + 1) + nT], nCf = offs[3 * (size_t)(nT + 1) + nT];
    if (nTp > 0x7fffffffll || nTf > 0x7fffffffll || nFv > 0x7fffffffll || nCf > 0x7fffffffll) return 1;      // (the host build words the error)
    int *tpIds = D.get<int>((size_t)nTp), *tfIds = D.get<int>((size_t)nTf), *meta = D.get<int>(12 * (size_t)nT);
    uint16_t *faceVerts = D.get<uint16_t>((size_t)nFv), *cellFaces = D.get<uint16_t>((size_t)nCf);
    if (!tpIds || !tfIds || !meta || !faceVerts || !cellFaces) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemsetAsync(cellFaces, 0xFF, (size_t)nCf * 2, st));
    hipLaunchKernelGGL(k_tl_geom<true>, dim3(nT), dim3(kT), 0, st, in, sz, GeomOut{oA, oB, oC, oD, tpIds, tfIds, faceVerts, cellFaces, meta});
    // what the host reads: offsets, widths, flags, maxima, the face ids (the face-angle filter's positions come from them)
    std::vector<int> hN((size_t)nT), hF((size_t)nT), hFw((size_t)nT), hCw((size_t)nT), hFl((size_t)nT);
    gt.tfIds.resize((size_t)nTf);
    TL_OK(hipMemcpyAsync(hN.data(), nPts, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hF.data(), nFcs, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hFw.data(), fw, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hCw.data(), cw, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hFl.data(), flags, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(gt.tfIds.data(), tfIds, (size_t)nTf * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    TL_OK(hipGetLastError());
    gt.tpOff.resize((size_t)nT + 1); gt.tfOff.resize((size_t)nT + 1); gt.fvBase.resize((size_t)nT); gt.cfBase.resize((size_t)nT);
    gt.fvWidth.resize((size_t)nT); gt.cfWidth.resize((size_t)nT); gt.tileFlags.resize((size_t)nT);
    gt.maxPoints = gt.maxFaces = 0;
    for (int i = 0; i <= nT; ++i) { gt.tpOff[(size_t)i] = (int32_t)offs[(size_t)i]; gt.tfOff[(size_t)i] = (int32_t)offs[(size_t)(nT + 1) + i]; }
    for (int i = 0; i < nT; ++i) {
        gt.fvBase[(size_t)i] = (int32_t)offs[2 * (size_t)(nT + 1) + i]; gt.cfBase[(size_t)i] = (int32_t)offs[3 * (size_t)(nT + 1) + i];
        gt.fvWidth[(size_t)i] = (uint8_t)hFw[(size_t)i]; gt.cfWidth[(size_t)i] = (uint8_t)hCw[(size_t)i]; gt.tileFlags[(size_t)i] = (uint8_t)hFl[(size_t)i];
        gt.maxPoints = std::max(gt.maxPoints, hN[(size_t)i]); gt.maxFaces = std::max(gt.maxFaces, hF[(size_t)i]);
    }
    gt.tpIds.clear(); gt.faceVerts.clear(); gt.cellFaces.clear();      // (device only: tpIds' length is tpOff.back())
    auto give = [&](GeomTilesDev::Arr& a, void* p, size_t bytes) { a.p = p; a.bytes = std::max<size_t>(bytes, 1); D.release(p); };
    give(out.cellOrder, dOrder, (size_t)nCells * 4); give(out.cellBeg, dBeg, ((size_t)nT + 1) * 4);
    give(out.tpIds, tpIds, (size_t)nTp * 4); give(out.tfIds, tfIds, (size_t)nTf * 4);
    give(out.faceVerts, faceVerts, (size_t)nFv * 2); give(out.cellFaces, cellFaces, (size_t)nCf * 2); give(out.meta, meta, 12 * (size_t)nT * 4);
    out.valid = true;
    return 0;
}

// et.order / et.edgeBeg / et.nTiles / et.threads stand (host: EdgeTiles::buildBoundaries).  Return values as buildGeomTablesOnDevice.
int buildEdgeTablesOnDevice(EdgeTiles& et, const DeviceTopologyArrays& td, int32_t nEdges, int device, EdgeTilesDev& out, std::string& why) {
    if (!td.valid || et.threads != kT || et.nTiles <= 0) return 1;
    TL_OK(hipSetDevice(device));
    hipStream_t st = nullptr;
    TL_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};
    const int nT = et.nTiles;
    DevBuf D;
    int *dOrder = D.get<int>((size_t)nEdges), *dBeg = D.get<int>((size_t)nT + 1);
    int *nPts = D.get<int>((size_t)nT), *nFcs = D.get<int>((size_t)nT), *nCls = D.get<int>((size_t)nT), *wf = D.get<int>((size_t)nT), *wc = D.get<int>((size_t)nT), *bad = D.get<int>(1);
    long long* terms = D.get<long long>(10 * ((size_t)nT + 1));
    if (!dOrder || !dBeg || !nPts || !nFcs || !nCls || !wf || !wc || !bad || !terms) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemcpyAsync(dOrder, et.order.data(), (size_t)nEdges * 4, hipMemcpyHostToDevice, st));
    TL_OK(hipMemcpyAsync(dBeg, et.edgeBeg.data(), ((size_t)nT + 1) * 4, hipMemcpyHostToDevice, st));
    TL_OK(hipMemsetAsync(bad, 0, 4, st));
    EdgeIn in{dOrder, dBeg, nT, (const int*)td.edges.p, (const int*)td.efOff.p, (const int*)td.efFace.p, (const int*)td.ecOff.p, (const int*)td.ecCell.p,
              (const int*)td.ringFace.p, (const int*)td.ringCell.p, (const uint8_t*)td.edgeRingOk.p};
    EdgeSizes sz{nPts, nFcs, nCls, wf, wc, bad};
    hipLaunchKernelGGL(k_tl_edge<false>, dim3(nT), dim3(kT), 0, st, in, sz, EdgeOut{});
    const size_t S = (size_t)nT + 1;
    TL_OK(hipMemsetAsync(terms, 0, 10 * S * 8, st));
    hipLaunchKernelGGL(k_tl_edgeTerms, dim3((nT + kT - 1) / kT), dim3(kT), 0, st, nT, sz, terms, terms + S, terms + 2 * S, terms + 3 * S, terms + 4 * S);
    long long* offsD = terms + 5 * S;
    size_t tempBytes = 0;
    (void)rocprim::exclusive_scan(nullptr, tempBytes, terms, offsD, 0ll, S, rocprim::plus<long long>(), st);
    void* temp = D.get<char>(tempBytes + 256);
    if (!temp) { why = "device allocation failed"; return 2; }
    for (int q = 0; q < 5; ++q) { size_t b = tempBytes; TL_OK(rocprim::exclusive_scan(temp, b, terms + (size_t)q * S, offsD + (size_t)q * S, 0ll, S, rocprim::plus<long long>(), st)); }
    std::vector<long long> offs(5 * S);
    int hbad = 0;
    TL_OK(hipMemcpyAsync(offs.data(), offsD, offs.size() * 8, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    if (hbad) return 1;
    const long long nTp = offs[(size_t)nT], nTf = offs[S + nT], nTc = offs[2 * S + nT], nEf = offs[3 * S + nT], nEc = offs[4 * S + nT];
    if (nTp > 0x7fffffffll || nTf > 0x7fffffffll || nTc > 0x7fffffffll || nEf > 0x7fffffffll || nEc > 0x7fffffffll) return 1;
    int *tpIds = D.get<int>((size_t)nTp), *tfIds = D.get<int>((size_t)nTf), *tcIds = D.get<int>((size_t)nTc), *meta = D.get<int>(12 * (size_t)nT);
    uint16_t *epLoc = D.get<uint16_t>(2 * (size_t)nEdges), *efEll = D.get<uint16_t>((size_t)nEf), *ecEll = D.get<uint16_t>((size_t)nEc);
    if (!tpIds || !tfIds || !tcIds || !meta || !epLoc || !efEll || !ecEll) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemsetAsync(efEll, 0xFF, (size_t)nEf * 2, st));
    TL_OK(hipMemsetAsync(ecEll, 0xFF, (size_t)nEc * 2, st));
    TL_OK(hipMemsetAsync(epLoc, 0, 2 * (size_t)nEdges * 2, st));
    hipLaunchKernelGGL(k_tl_edge<true>, dim3(nT), dim3(kT), 0, st, in, sz, EdgeOut{offsD, offsD + S, offsD + 2 * S, offsD + 3 * S, offsD + 4 * S, tpIds, tfIds, tcIds, epLoc, efEll, ecEll, meta});
    std::vector<int> hP((size_t)nT), hF((size_t)nT), hC((size_t)nT), hWf((size_t)nT), hWc((size_t)nT);
    TL_OK(hipMemcpyAsync(hP.data(), nPts, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hF.data(), nFcs, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hC.data(), nCls, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hWf.data(), wf, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipMemcpyAsync(hWc.data(), wc, (size_t)nT * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    TL_OK(hipGetLastError());
    et.tpOff.resize(S); et.tfOff.resize(S); et.tcOff.resize(S); et.efBase.resize((size_t)nT); et.ecBase.resize((size_t)nT); et.efWidth.resize((size_t)nT); et.ecWidth.resize((size_t)nT);
    et.maxPoints = et.maxFaces = et.maxCells = 0;
    for (size_t i = 0; i < S; ++i) { et.tpOff[i] = (int32_t)offs[i]; et.tfOff[i] = (int32_t)offs[S + i]; et.tcOff[i] = (int32_t)offs[2 * S + i]; }
    for (int i = 0; i < nT; ++i) {
        et.efBase[(size_t)i] = (int32_t)offs[3 * S + i]; et.ecBase[(size_t)i] = (int32_t)offs[4 * S + i];
        et.efWidth[(size_t)i] = (uint8_t)hWf[(size_t)i]; et.ecWidth[(size_t)i] = (uint8_t)hWc[(size_t)i];
        et.maxPoints = std::max(et.maxPoints, hP[(size_t)i]); et.maxFaces = std::max(et.maxFaces, hF[(size_t)i]); et.maxCells = std::max(et.maxCells, hC[(size_t)i]);
    }
    et.tpIds.clear(); et.tfIds.clear(); et.tcIds.clear(); et.epLoc.clear(); et.efEll.clear(); et.ecEll.clear();
    auto give = [&](EdgeTilesDev::Arr& a, void* p, size_t bytes) { a.p = p; a.bytes = std::max<size_t>(bytes, 1); D.release(p); };
    give(out.order, dOrder, (size_t)nEdges * 4); give(out.edgeBeg, dBeg, S * 4);
    give(out.tpIds, tpIds, (size_t)nTp * 4); give(out.tfIds, tfIds, (size_t)nTf * 4); give(out.tcIds, tcIds, (size_t)nTc * 4);
    give(out.epLoc, epLoc, 2 * (size_t)nEdges * 2); give(out.efEll, efEll, (size_t)nEf * 2); give(out.ecEll, ecEll, (size_t)nEc * 2); give(out.meta, meta, 12 * (size_t)nT * 4);
    out.nTf = nTf;
    out.valid = true;
    return 0;
}

// The corner chains of ALL points, started ahead of the tables (they need the device addressing only, not the tile boundaries):
// 0.2 s of kernel for 10 M cells -- a sequential graph walk per thread on arrays in scratch memory -- that then runs beside the
// host's boundary pass.  buildSmoothTablesOnDevice takes the result over (and frees it); releaseCornerChains for a caller that
// never gets there.
int startCornerChains(const DeviceTopologyArrays& td, int32_t nPoints, int device, CornerChains& c, std::string& why) {
    if (!td.valid || nPoints <= 0) return 1;
    TL_OK(hipSetDevice(device));
    const size_t nPf = td.pfPrev.bytes / 4;
    hipStream_t st = nullptr;
    TL_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *a = nullptr, *b = nullptr, *bad = nullptr, *big = nullptr;
    if (hipMalloc((void**)&a, std::max<size_t>(nPf, 1) * 4) != hipSuccess || hipMalloc((void**)&b, std::max<size_t>(nPf, 1) * 4) != hipSuccess || hipMalloc((void**)&bad, 4) != hipSuccess ||
        hipMalloc((void**)&big, ((size_t)nPoints + 1) * 4) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (bad) (void)hipFree(bad);
        (void)hipStreamDestroy(st);
        why = "device allocation failed";
        return 2;
    }
    c.chPrev = a; c.chNext = b; c.bad = bad; c.big = big; c.stream = (void*)st; c.nPoints = nPoints;
    (void)hipMemsetAsync(bad, 0, 4, st);
    launchChains(st, nPoints, nullptr, (const int*)td.pfOff.p, (const int*)td.pfPrev.p, (const int*)td.pfNext.p, a, b, big, bad);
    c.started = true;
    return 0;
}
void releaseCornerChains(CornerChains& c) {
    if (!c.started) return;
    (void)hipStreamSynchronize((hipStream_t)c.stream);
    (void)hipStreamDestroy((hipStream_t)c.stream);
    (void)hipFree(c.chPrev); (void)hipFree(c.chNext); (void)hipFree(c.bad);
    if (c.big) (void)hipFree(c.big);
    c = CornerChains();
}

// st.order / st.ptBeg / st.nTiles / st.threads stand (host: SmoothTiles::buildBoundaries).  Return values as buildGeomTablesOnDevice.
int buildSmoothTablesOnDevice(SmoothTiles& st, const DeviceTopologyArrays& td, int32_t nPoints, int32_t maxPointPoints, const uint8_t* isInternal, int device,
                              SmoothTilesDev& out, std::string& why, CornerChains* chains) {
    struct ChainGuard { CornerChains* c; ~ChainGuard() { if (c) releaseCornerChains(*c); } } chainGuard{chains};
    if (!td.valid || st.threads != kT || st.nTiles <= 0 || st.order.empty()) return 1;
    const bool haveChains = chains && chains->started && chains->nPoints == nPoints;
    const int nPos = (int)st.order.size();      // (all points, or the subset the tiles are made of)
    TL_OK(hipSetDevice(device));
    hipStream_t sm = nullptr;
    TL_OK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{sm};
    const int nT = st.nTiles;
    const size_t S = (size_t)nT + 1, nPf = td.pfPrev.bytes / 4;
    DevBuf D;
    int *dOrder = D.get<int>((size_t)nPos), *dBeg = D.get<int>(S);
    int *chPrev = haveChains ? chains->chPrev : D.get<int>(nPf), *chNext = haveChains ? chains->chNext : D.get<int>(nPf);
    uint8_t* dInt = D.get<uint8_t>((size_t)nPoints);
    int *nCl = D.get<int>((size_t)nT), *nPt = D.get<int>((size_t)nT), *wc = D.get<int>((size_t)nT), *wn = D.get<int>((size_t)nT), *wf = D.get<int>((size_t)nT), *bad = D.get<int>(1);
    long long* terms = D.get<long long>(10 * S);
    if (!dOrder || !dBeg || !chPrev || !chNext || !dInt || !nCl || !nPt || !wc || !wn || !wf || !bad || !terms) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemsetAsync(bad, 0, 4, sm));
    TL_OK(hipMemcpyAsync(dOrder, st.order.data(), (size_t)nPos * 4, hipMemcpyHostToDevice, sm));
    if (haveChains) {      // (started ahead: wait for it, and take its verdict on the corner counts)
        TL_OK(hipStreamSynchronize((hipStream_t)chains->stream));
        int cb = 0;
        TL_OK(hipMemcpy(&cb, chains->bad, 4, hipMemcpyDeviceToHost));
        if (cb) return 1;
    } else {
        int* big = D.get<int>((size_t)nPos + 1);
        if (!big) { why = "device allocation failed"; return 2; }
        launchChains(sm, nPos, nPos == nPoints ? (const int*)nullptr : (const int*)dOrder, (const int*)td.pfOff.p, (const int*)td.pfPrev.p, (const int*)td.pfNext.p, chPrev, chNext, big, bad);
    }
    TL_OK(hipMemcpyAsync(dBeg, st.ptBeg.data(), S * 4, hipMemcpyHostToDevice, sm));
    TL_OK(hipMemcpyAsync(dInt, isInternal, (size_t)nPoints, hipMemcpyHostToDevice, sm));
    SmoothIn in{dOrder, dBeg, nT, maxPointPoints <= 16 ? 1 : 0, (const int*)td.pcOff.p, (const int*)td.pcVal.p, (const int*)td.ppOff.p, (const int*)td.ppPt.p,
                (const int*)td.pfOff.p, chPrev, chNext, dInt};
    SmoothSizes sz{nCl, nPt, wc, wn, wf, bad};
    hipLaunchKernelGGL(k_tl_smooth<false>, dim3(nT), dim3(kT), 0, sm, in, sz, SmoothOut{});
    TL_OK(hipMemsetAsync(terms, 0, 10 * S * 8, sm));
    hipLaunchKernelGGL(k_tl_smoothTerms, dim3((nT + kT - 1) / kT), dim3(kT), 0, sm, nT, sz, terms, terms + S, terms + 2 * S, terms + 3 * S, terms + 4 * S);
    long long* offsD = terms + 5 * S;
    size_t tempBytes = 0;
    (void)rocprim::exclusive_scan(nullptr, tempBytes, terms, offsD, 0ll, S, rocprim::plus<long long>(), sm);
    void* temp = D.get<char>(tempBytes + 256);
    if (!temp) { why = "device allocation failed"; return 2; }
    for (int q = 0; q < 5; ++q) { size_t b = tempBytes; TL_OK(rocprim::exclusive_scan(temp, b, terms + (size_t)q * S, offsD + (size_t)q * S, 0ll, S, rocprim::plus<long long>(), sm)); }
    std::vector<long long> offs(5 * S);
    int hbad = 0;
    TL_OK(hipMemcpyAsync(offs.data(), offsD, offs.size() * 8, hipMemcpyDeviceToHost, sm));
    TL_OK(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipStreamSynchronize(sm));
    if (hbad) return 1;
    const long long nTc = offs[(size_t)nT], nTn = offs[S + nT], nPc = offs[2 * S + nT], nPp = offs[3 * S + nT], nPfE = offs[4 * S + nT];
    if (nTc > 0x7fffffffll || nTn > 0x7fffffffll || nPc > 0x7fffffffll || nPp > 0x7fffffffll || nPfE > 0x7fffffffll) return 1;
    int *tcIds = D.get<int>((size_t)nTc), *tnIds = D.get<int>((size_t)nTn), *meta = D.get<int>(12 * (size_t)nT);
    uint16_t *selfLoc = D.get<uint16_t>((size_t)nPos), *pcEll = D.get<uint16_t>((size_t)nPc), *ppEll = D.get<uint16_t>((size_t)nPp), *pairEll = D.get<uint16_t>((size_t)nPp),
             *pfEll = D.get<uint16_t>((size_t)nPfE);
    if (!tcIds || !tnIds || !meta || !selfLoc || !pcEll || !ppEll || !pairEll || !pfEll) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemsetAsync(pcEll, 0xFF, (size_t)nPc * 2, sm));
    TL_OK(hipMemsetAsync(ppEll, 0xFF, (size_t)nPp * 2, sm));
    TL_OK(hipMemsetAsync(pfEll, 0xFF, (size_t)nPfE * 2, sm));
    TL_OK(hipMemsetAsync(pairEll, 0, (size_t)nPp * 2, sm));
    TL_OK(hipMemsetAsync(selfLoc, 0, (size_t)nPos * 2, sm));
    hipLaunchKernelGGL(k_tl_smooth<true>, dim3(nT), dim3(kT), 0, sm, in, sz,
                       SmoothOut{offsD, offsD + S, offsD + 2 * S, offsD + 3 * S, offsD + 4 * S, tcIds, tnIds, selfLoc, pcEll, ppEll, pairEll, pfEll, meta});
    std::vector<int> hC((size_t)nT), hN((size_t)nT), hWc((size_t)nT), hWn((size_t)nT), hWf((size_t)nT);
    TL_OK(hipMemcpyAsync(hC.data(), nCl, (size_t)nT * 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipMemcpyAsync(hN.data(), nPt, (size_t)nT * 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipMemcpyAsync(hWc.data(), wc, (size_t)nT * 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipMemcpyAsync(hWn.data(), wn, (size_t)nT * 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipMemcpyAsync(hWf.data(), wf, (size_t)nT * 4, hipMemcpyDeviceToHost, sm));
    TL_OK(hipStreamSynchronize(sm));
    TL_OK(hipGetLastError());
    st.tcOff.resize(S); st.tnOff.resize(S);
    st.pcBase.resize((size_t)nT); st.ppBase.resize((size_t)nT); st.pfBase.resize((size_t)nT); st.pcWidth.resize((size_t)nT); st.ppWidth.resize((size_t)nT); st.pfWidth.resize((size_t)nT);
    st.maxCells = st.maxPoints = 0;
    for (size_t i = 0; i < S; ++i) { st.tcOff[i] = (int32_t)offs[i]; st.tnOff[i] = (int32_t)offs[S + i]; }
    for (int i = 0; i < nT; ++i) {
        st.pcBase[(size_t)i] = (int32_t)offs[2 * S + i]; st.ppBase[(size_t)i] = (int32_t)offs[3 * S + i]; st.pfBase[(size_t)i] = (int32_t)offs[4 * S + i];
        st.pcWidth[(size_t)i] = (uint8_t)hWc[(size_t)i]; st.ppWidth[(size_t)i] = (uint8_t)hWn[(size_t)i]; st.pfWidth[(size_t)i] = (uint8_t)hWf[(size_t)i];
        st.maxCells = std::max(st.maxCells, hC[(size_t)i]); st.maxPoints = std::max(st.maxPoints, hN[(size_t)i]);
    }
    st.tcIds.clear(); st.tnIds.clear(); st.selfLoc.clear(); st.pcEll.clear(); st.ppEll.clear(); st.pairEll.clear(); st.pfEll.clear();
    auto give = [&](SmoothTilesDev::Arr& a, void* p, size_t bytes) { a.p = p; a.bytes = std::max<size_t>(bytes, 1); D.release(p); };
    give(out.order, dOrder, (size_t)nPos * 4); give(out.ptBeg, dBeg, S * 4); give(out.tcIds, tcIds, (size_t)nTc * 4); give(out.tnIds, tnIds, (size_t)nTn * 4);
    give(out.selfLoc, selfLoc, (size_t)nPos * 2); give(out.pcEll, pcEll, (size_t)nPc * 2); give(out.ppEll, ppEll, (size_t)nPp * 2); give(out.pairEll, pairEll, (size_t)nPp * 2);
    give(out.pfEll, pfEll, (size_t)nPfE * 2); give(out.meta, meta, 12 * (size_t)nT * 4);
    out.valid = true;
    return 0;
}

// order = the cells sorted along the Z-curve of their vertex averages, as tiles.cpp's mortonOrder gives it.  0 done; 1 not handled;
// 2 a HIP error
int cellMortonOrderOnDevice(const DeviceTopologyArrays& td, int32_t nCells, int32_t nPoints, const double* points, int device, std::vector<int32_t>& order, std::string& why) {
    if (!td.valid || nCells <= 0 || nPoints <= 0) return 1;
    TL_OK(hipSetDevice(device));
    hipStream_t st = nullptr;
    TL_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};
    const int nB = (nCells + kT - 1) / kT;
    DevBuf D;
    double *dPts = D.get<double>(3 * (size_t)nPoints), *cc = D.get<double>(3 * (size_t)nCells), *part = D.get<double>(6 * (size_t)nB);
    u64 *kA = D.get<u64>((size_t)nCells), *kB = D.get<u64>((size_t)nCells);
    int *iA = D.get<int>((size_t)nCells), *iB = D.get<int>((size_t)nCells);
    if (!dPts || !cc || !part || !kA || !kB || !iA || !iB) { why = "device allocation failed"; return 2; }
    TL_OK(hipMemcpyAsync(dPts, points, 3 * (size_t)nPoints * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tl_cellAvg, dim3(nB), dim3(kT), 0, st, nCells, (const int*)td.cfOff.p, (const int*)td.cfVal.p, (const int*)td.faceOff.p, (const int*)td.facePts.p, dPts, cc, part);
    std::vector<double> hp(6 * (size_t)nB);
    TL_OK(hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < nB; ++b)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], hp[6 * (size_t)b + a]); hi[a] = std::max(hi[a], hp[6 * (size_t)b + 3 + a]); }
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, hi[a] - lo[a]);
    const double scale = ext > 0.0 ? 2097151.0 / ext : 0.0;
    hipLaunchKernelGGL(k_tl_mortonKeys, dim3(nB), dim3(kT), 0, st, nCells, cc, lo[0], lo[1], lo[2], scale, kA, iA);
    size_t tempBytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tempBytes, kA, kB, iA, iB, (size_t)nCells, 0, 63, st);
    void* temp = D.get<char>(tempBytes + 256);
    if (!temp) { why = "device allocation failed"; return 2; }
    TL_OK(rocprim::radix_sort_pairs(temp, tempBytes, kA, kB, iA, iB, (size_t)nCells, 0, 63, st));
    order.resize((size_t)nCells);
    TL_OK(hipMemcpyAsync(order.data(), iB, (size_t)nCells * 4, hipMemcpyDeviceToHost, st));
    TL_OK(hipStreamSynchronize(st));
    TL_OK(hipGetLastError());
    return 0;
}

// ids[k] = position of face ids[k] in the geometry tiles' face lists (device arrays); 0 ok, 2 HIP error
int remapEdgeFaceIdsOnDevice(const int* geomTfIds, long long nGeomTf, int32_t nFaces, int* edgeTfIds, long long nEdgeTf, int device, std::string& why) {
    TL_OK(hipSetDevice(device));
    int* facePos = nullptr;
    TL_OK(hipMalloc((void**)&facePos, std::max<size_t>((size_t)nFaces, 1) * 4));
    struct Free { int* p; ~Free() { (void)hipFree(p); } } guard{facePos};
    TL_OK(hipMemset(facePos, 0xFF, (size_t)nFaces * 4));
    hipLaunchKernelGGL(k_tl_facePos, dim3((unsigned)((nGeomTf + kT - 1) / kT)), dim3(kT), 0, nullptr, nGeomTf, geomTfIds, facePos);
    hipLaunchKernelGGL(k_tl_remap, dim3((unsigned)((nEdgeTf + kT - 1) / kT)), dim3(kT), 0, nullptr, nEdgeTf, edgeTfIds, facePos);
    TL_OK(hipDeviceSynchronize());
    TL_OK(hipGetLastError());
    return 0;
}

}  // namespace smgpu
