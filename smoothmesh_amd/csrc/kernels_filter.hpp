// kernels_filter.hpp -- floating-point filters for the two angle evaluators.
//
// Both evaluators end in threshold tests (edge angle < minAngle, SM.C:923; face angles outside
// (minAngle, maxAngle), SM.C:1367) and on a decent mesh almost every point / edge is far from the
// thresholds, yet the exact f64 evaluation costs ~400 (edge, cell) / ~380 (point, face) FP64
// instructions of sqrt / div / acos.  As in exact geometric predicates, a cheap conservative filter
// runs first: the same quantity in f32 (difference vectors are formed in f64 first, so large
// coordinates cannot cancel), with an error margin orders of magnitude above the f32 error bound.
// Only elements the filter cannot decide are evaluated exactly, by the unchanged f64 kernels.
// The filter never decides a borderline case, so frozen sets are identical to the unfiltered run.
//
// Error budget (angles in radians): unit vectors from f64 differences rounded to f32: <= 2e-7 per
// component; a projection is trusted only if the in-plane part keeps > 1 % of the vector (else
// UNSURE), so cancellation amplifies by < 100: < 2e-5; acos amplifies by 1/sqrt(1-x^2) <= 224 at the
// clamp +-0.99999: < 5e-3 worst case at the clamp, < 1e-4 for |cos| < 0.999.  The margins below
// (kFaMargin on angle sums, kEaMargin on cosines) are applied together with a guard that sends every
// near-clamp cosine to the exact path.
//
// Face angles, round 4: the test "small + margin < acos(a) + acos(b) < large - margin" is made on the COSINE of the sum,
//   c = a b - sqrt((1 - a^2)(1 - b^2)),
// against thresholds the host prepares in f64 (Prm::faCosLo = cos(small) - M, Prm::faCosHi = cos(large) + M, smgpu.hip:makePrm)
// -- two acosf (19 vector instructions each) become one v_sqrt_f32 and four multiply-adds, a fifth of the kernel's
// instructions.  cos is 1-Lipschitz, so an error of the sum bounds the error of c: the same margin M = kFaMargin (+ 2e-5 for the
// rounding of the formula itself: 1 - a^2 >= 2e-3 behind the near-clamp guard) keeps the same distance from the f32 error bound.
// cos decreases on [0, pi] only, so "sum < large" is accepted only with a + b > kFaSumMin (true a + b > 0 <=> the sum is below
// pi); a sum above pi has c rising again and fails "c < faCosLo" at worst -- UNSURE, never wrongly GOOD.
#pragma once
#include "kernels.hpp"
#include "kernels_tiled.hpp"

namespace smgpu {

constexpr float kFaMargin = 2.0e-3f;     // rad, on acos(a) + acos(b): applied to the cosine of the sum (Prm::faCosLo / faCosHi)
constexpr float kFaSumMin = 2.0e-3f;     // a + b above this: acos(a) + acos(b) < pi for sure
constexpr float kEaMargin = 1.0e-3f;     // on a clamped cosine
constexpr float kNearClamp = 0.999f;     // |cos| above this goes to the exact path

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(const V3& v) { F3 r; r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; return r; }
// (explicit fused multiply-adds: the library is built with -ffp-contract=off for the reference's f64 arithmetic, which left the
// filters' f32 dot products as three multiplies and two adds -- they are estimates behind margins, any rounding serves, and the
// filter kernels are bound by vector issue)
__device__ __forceinline__ float fdot(const F3& a, const F3& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// a - t b
__device__ __forceinline__ F3 fsubScaled(const F3& a, float t, const F3& b) { F3 r; r.x = fmaf(-t, b.x, a.x); r.y = fmaf(-t, b.y, a.y); r.z = fmaf(-t, b.z, a.z); return r; }
__device__ __forceinline__ F3 fscale(const F3& a, float s) { F3 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
__device__ __forceinline__ F3 funit(const F3& a, bool& ok) {
    const float n2 = fdot(a, a);
    ok = ok && (n2 > 1.0e-30f) && (n2 < 1.0e30f);
    return fscale(a, rsqrtf(n2));
}

// acos(a) + acos(b) inside (small + margin, large - margin) for sure, decided on the cosine of the sum (file header)
__device__ __forceinline__ bool faSumInside(float a, float b, const Prm& prm) {
    const float c = fmaf(a, b, -__builtin_amdgcn_sqrtf(fmaf(-a, a, 1.0f) * fmaf(-b, b, 1.0f)));
    return (c < prm.faCosLo) && (c > prm.faCosHi) && (a + b > kFaSumMin);
}

// ---- face angles: per edge GOOD (0) / UNSURE (1) -------------------------------------------------------
// GOOD = every cell angle of the edge lies inside (small + margin, large - margin) for sure.
// An UNSURE edge marks both end points in faMaybe with the iteration's tag (State::faGen; like faActive): a point may be
// outside the good range only if one of its edges is UNSURE; every other point is inside for sure (SM.C:1367-1369).
__device__ __forceinline__ void markUnsureEdge(const State& s, const int* edges, int e, uint8_t* faMaybe) {
    faMaybe[edges[2 * e]] = s.faGen;    // racing writers all store this iteration's tag
    faMaybe[edges[2 * e + 1]] = s.faGen;
    atomicAdd(&s.acc->nFaMaybe, 1);     // rare on a decent mesh
}

__global__ void __launch_bounds__(kBlock) k_fa_edges_filter(MeshView m, State s, Prm prm, uint8_t* faMaybe) {
    if (s.acc->stop) return;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= m.nEdges) return;
    uint8_t flag = 1;
    if (m.edgeRingOk[e]) {
        const V3 e0 = ldv(s.ptsCur, m.edges[2 * e]), e1 = ldv(s.ptsCur, m.edges[2 * e + 1]);
        const V3 cC = 0.5 * (e0 + e1);
        bool ok = true;
        const F3 eV = funit(f3(e1 - e0), ok);
        // unit vector of the part of (c - cC) perpendicular to the edge (SM.C:1189-1196 up to rounding)
        auto project = [&](const V3& c) -> F3 {
            const F3 d = f3(c - cC);
            const float t = fdot(d, eV);
            const F3 w = fsubScaled(d, t, eV);
            const float w2 = fdot(w, w), d2 = fdot(d, d);
            ok = ok && (w2 > 1.0e-4f * d2);          // in-plane part keeps > 1 % of the vector
            return funit(w, ok);
        };
        const int fb = m.efOff[e], nf = m.efOff[e + 1] - fb;
        const int cb = m.ecOff[e], nc = m.ecOff[e + 1] - cb;
        const F3 first = project(ldv(s.fAvg, m.ringFace[fb]));
        F3 prev = first;
        bool inside = true;
        for (int i = 0; i < nc; ++i) {
            const F3 next = (i + 1 < nf) ? project(ldv(s.fAvg, m.ringFace[fb + i + 1])) : first;
            const F3 cV = project(ldv(s.cellCtr, m.ringCell[cb + i]));
            const float a = fdot(prev, cV), b = fdot(cV, next);
            ok = ok && (fabsf(a) < kNearClamp) && (fabsf(b) < kNearClamp);
            inside = inside && faSumInside(a, b, prm);      // NaN -> false -> UNSURE
            prev = next;
        }
        if (ok && inside && nc > 0) flag = 0;
    }
    if (flag) markUnsureEdge(s, m.edges, e, faMaybe);
}

// ---- edge angles: per point "may freeze" ----------------------------------------------------------------
// A point is frozen iff minN < small && minN < minC (SM.C:923).  If every new-angle cosine is below
// cos(small) - margin, minN > small for sure and the exact kernel can skip the point.
__global__ void __launch_bounds__(kBlock) k_edge_angle_filter(MeshView m, State s, Prm prm, float cosSmall, uint8_t* eaMaybe) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= m.nPoints) return;
    if (s.frozen[p]) { eaMaybe[p] = 0; return; }
    const V3 np0 = ldv(s.prop, p);
    const float thr = cosSmall - kEaMargin;
    bool ok = true, below = true;
    const int b = m.pfOff[p], e = m.pfOff[p + 1];
    for (int k = b; k < e; ++k) {
        const int a1 = m.pfPrev[k], a2 = m.pfNext[k];
        const F3 uc1 = funit(f3(ldv(s.ptsCur, a1) - np0), ok), uc2 = funit(f3(ldv(s.ptsCur, a2) - np0), ok);
        const F3 un1 = funit(f3(ldv(s.prop, a1) - np0), ok), un2 = funit(f3(ldv(s.prop, a2) - np0), ok);
        const float c0 = fdot(uc1, uc2), c1 = fdot(un1, un2), c2 = fdot(uc1, un2), c3 = fdot(un1, uc2);
        below = below && (c0 < thr) && (c1 < thr) && (c2 < thr) && (c3 < thr);   // NaN -> false -> exact path
    }
    eaMaybe[p] = (ok && below) ? 0 : 1;
    if (!(ok && below)) atomicAdd(&s.acc->nEaMaybe, 1);
}

// The same filter on the smoothing tiles: current and proposed coordinates of the tile's points and their
// neighbours are staged in LDS once (the per-point form above gathers each neighbour 8 times from global
// memory and is bound by those gathers, not by arithmetic); corner pairs come from the pfEll table.
//
// The staged coordinates are f32 offsets from a tile origin O (the tile's first staged point, subtracted in f64): a tile
// spans ~10 edge lengths, so an offset carries an absolute error <= 2^-24 R (R = largest offset of the tile) and a difference
// of two of them <= 1.2e-7 R.  A vector shorter than sqrt(kRelGuard) R = 1.4e-3 R sends the point to the exact kernel;
// every other unit vector is good to 8.5e-5 and a cosine to 1.7e-4, a sixth of kEaMargin.  (R is the whole tile's: a tile
// whose points straddle a jump of the Morton order has a large R, and a tighter guard sent many of its points to the exact kernel.)  With the offsets in f32 the (current,
// proposed) pair of each neighbour goes through packed two-wide f32 instructions (v_pk_mul_f32 / v_pk_add_f32): the kernel is
// bound by VALU issue, and the f64 subtract + convert of every vector was a quarter of it.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }      // v_pk_fma_f32 (see fdot)
constexpr float kRelGuard = 2.0e-6f;

// block-wide maximum of a non-negative float (all T threads call it); red: T/64 floats of LDS
template <int T>
__device__ __forceinline__ float blockMaxF(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int i = 1; i < T / 64; ++i) m = red[i] > m ? red[i] : m;
    return m;
}

template <int T>
__global__ void __launch_bounds__(T) k_ea_filter_tile(MeshView m, State s, SmoothTileView g, float cosSmall, uint8_t* eaMaybe,
                                                       int nLaunch, int xcdMap) {
    if (s.acc->stop) return;
    const int li = launchTile(nLaunch, xcdMap);
    if (li < 0) return;
    extern __shared__ double ldsRaw[];
    float* lds = reinterpret_cast<float*>(ldsRaw);
    float* cx = lds;                float* cy = cx + g.maxPoints; float* cz = cy + g.maxPoints;
    float* nx = cz + g.maxPoints;   float* ny = nx + g.maxPoints; float* nz = ny + g.maxPoints;
    float* red = nz + g.maxPoints;
    const int tile = li, tid = threadIdx.x;
    // prologue in two dependent round trips, as in smoothStage: (1) id list, point id / slot, the first kEaPre chunks of the
    // corner row (lanes without a point read lane 0's; chunks past the row's width read chunk 0); (2) current and proposed
    // coordinates, the point's frozen flag.  The tile's origin (first point of its list) comes through scalar loads.
    constexpr int kEaPre = 6;          // 12 faces per point (hexahedral meshes) = 24 entries
    const SmoothTileMeta tm = loadTileMeta(g, tile);
    const bool mine = tid < tm.nPts;
    const int wf4 = tm.pfWidth >> 2;
    const ushort4* row = reinterpret_cast<const ushort4*>(g.pfEll + tm.pfBase) + tid;
    const int n = tm.nNbrs;
    const int* ids = g.tnIds + tm.tnOff;
    typedef const __attribute__((address_space(4))) double* const_dbl_ptr;
    V3 O = v3(0, 0, 0);
    if (n > 0) {
        const int id0 = ((const_int_ptr)ids)[0];
        const_dbl_ptr o = (const_dbl_ptr)(s.ptsCur + 3 * (size_t)id0);
        O = v3(o[0], o[1], o[2]);
    }
    constexpr int R = 3;
    int id[R];
#pragma unroll
    for (int u = 0; u < R; ++u) { const int i = u * T + tid; id[u] = (i < n) ? ids[i] : -1; }
    const int lane = mine ? tid : 0;
    const int pi = tm.ptBeg + lane;
    int p = g.ptOrder[pi], selfL = g.selfLoc[pi];
    ushort4 qs[kEaPre];
    {
        const ushort4* rl = reinterpret_cast<const ushort4*>(g.pfEll + tm.pfBase) + lane;
#pragma unroll
        for (int c = 0; c < kEaPre; ++c) qs[c] = rl[(size_t)(c < wf4 ? c : 0) * T];     // (rows have at least one chunk: tiles.cpp)
    }
    float r2 = 0.f;
    {
        V3 va[R], vb[R];
#pragma unroll
        for (int u = 0; u < R; ++u) { const int j = (id[u] >= 0) ? id[u] : p; va[u] = ldv(s.ptsCur, j); vb[u] = ldv(s.prop, j); }   // (no branch)
        const uint8_t fz = s.frozen[p];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = u * T + tid;
            if (id[u] >= 0) {
                const F3 a = f3(va[u] - O), c = f3(vb[u] - O);
                cx[i] = a.x; cy[i] = a.y; cz[i] = a.z; nx[i] = c.x; ny[i] = c.y; nz[i] = c.z;
                const float aa = fdot(a, a), cc = fdot(c, c);
                r2 = aa > r2 ? aa : r2;
                r2 = cc > r2 ? cc : r2;
            }
        }
        for (int base = T * R; base < n; base += T * R) {        // tiles beyond the fixed number of rounds
            int jd[R];
#pragma unroll
            for (int u = 0; u < R; ++u) { const int i = base + u * T + tid; jd[u] = (i < n) ? ids[i] : -1; }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = base + u * T + tid;
                if (jd[u] >= 0) {
                    const F3 a = f3(ldv(s.ptsCur, jd[u]) - O), c = f3(ldv(s.prop, jd[u]) - O);
                    cx[i] = a.x; cy[i] = a.y; cz[i] = a.z; nx[i] = c.x; ny[i] = c.y; nz[i] = c.z;
                    const float aa = fdot(a, a), cc = fdot(c, c);
                    r2 = aa > r2 ? aa : r2;
                    r2 = cc > r2 ? cc : r2;
                }
            }
        }
        if (!mine) { p = 0; selfL = 0; }
        selfL |= (int)fz << 16;                                   // carried across the barrier with the slot
    }
    r2 = blockMaxF<T>(r2, red);        // (also the barrier behind the staging)
    if (!mine) return;
    if (selfL >> 16) { eaMaybe[p] = 0; return; }
    selfL &= 0xffff;
    const float p0x = nx[selfL], p0y = ny[selfL], p0z = nz[selfL];
    const float thr = cosSmall - kEaMargin;
    const float minN2 = kRelGuard * r2;
    bool ok = r2 < 1.0e30f, below = true;
    // The corner list of a point is ordered as chains (tiles.cpp, chainCorners): a corner usually starts at the vertex
    // the previous one ended at, whose unit vectors are still at hand.  A unit-vector pair = (towards the neighbour's current
    // position, towards its proposal), both from this point's PROPOSAL (SM.C:868-885), as the two halves of f2 registers.
    int heldId = -1;
    f2 hx = {0.f, 0.f}, hy = {0.f, 0.f}, hz = {0.f, 0.f};
#define SMGPU_EA_UNIT(UX, UY, UZ, A)                                                                              \
    {                                                                                                             \
        const f2 vx_ = {cx[(A)] - p0x, nx[(A)] - p0x}, vy_ = {cy[(A)] - p0y, ny[(A)] - p0y}, vz_ = {cz[(A)] - p0z, nz[(A)] - p0z}; \
        const f2 n2_ = pkfma(vx_, vx_, pkfma(vy_, vy_, vz_ * vz_));                                               \
        ok = ok && (n2_.x > minN2) && (n2_.y > minN2);            /* NaN -> false -> exact path */                 \
        const f2 rs_ = {__builtin_amdgcn_rsqf(n2_.x), __builtin_amdgcn_rsqf(n2_.y)};                               \
        UX = vx_ * rs_; UY = vy_ * rs_; UZ = vz_ * rs_;                                                           \
    }
#define SMGPU_EA_CORNER(A1, A2)                                                                                   \
    if ((A1) != kPad) {                                                                                           \
        if ((int)(A1) != heldId) SMGPU_EA_UNIT(hx, hy, hz, (A1))                                                  \
        f2 ux, uy, uz;                                                                                            \
        SMGPU_EA_UNIT(ux, uy, uz, (A2))                                                                           \
        const f2 d0 = pkfma(hx, ux, pkfma(hy, uy, hz * uz));     /* (x1 . x2, n1 . n2): nAngle0, nAngle1 */        \
        const f2 d1 = pkfma(hx, ux.yx, pkfma(hy, uy.yx, hz * uz.yx));   /* (x1 . n2, n1 . x2): nAngle2, nAngle3 */ \
        below = below && (d0.x < thr) && (d0.y < thr) && (d1.x < thr) && (d1.y < thr);                            \
        hx = ux; hy = uy; hz = uz; heldId = (int)(A2);                                                            \
    }
#pragma unroll
    for (int c = 0; c < kEaPre; ++c)
        if (c < wf4) { SMGPU_EA_CORNER(qs[c].x, qs[c].y) SMGPU_EA_CORNER(qs[c].z, qs[c].w) }
    for (int c = kEaPre; c < wf4; ++c) {
        const ushort4 q = row[(size_t)c * T];
        SMGPU_EA_CORNER(q.x, q.y) SMGPU_EA_CORNER(q.z, q.w)
    }
#undef SMGPU_EA_CORNER
#undef SMGPU_EA_UNIT
    eaMaybe[p] = (ok && below) ? 0 : 1;
    if (!(ok && below)) atomicAdd(&s.acc->nEaMaybe, 1);
}

struct EdgeTileView {
    const int* order; const int* edgeBeg;
    const int* tpOff; const int* tpIds; const int* tfOff; const int* tfIds; const int* tcOff; const int* tcIds;
    const uint16_t* epLoc;
    const int* efBase; const int* ecBase; const uint8_t* efWidth; const uint8_t* ecWidth;
    const uint16_t* efEll; const uint16_t* ecEll;
    const int* meta;             // per-tile scalars, one record per tile (EdgeTileMeta)
    int maxPoints, maxFaces, maxCells;
};
struct EdgeTileMeta { int edgeBeg, nEdges, tpOff, nPts, tfOff, nFaces, tcOff, nCells, efBase, efWidth, ecBase, ecWidth; };
constexpr int kEdgeMetaInts = 12;
__device__ __forceinline__ EdgeTileMeta loadTileMeta(const EdgeTileView& g, int tile) {
    const_int_ptr p = (const_int_ptr)(g.meta + (size_t)kEdgeMetaInts * tile);
    EdgeTileMeta t;
    t.edgeBeg = p[0]; t.nEdges = p[1]; t.tpOff = p[2]; t.nPts = p[3]; t.tfOff = p[4]; t.nFaces = p[5];
    t.tcOff = p[6]; t.nCells = p[7]; t.efBase = p[8]; t.efWidth = p[9]; t.ecBase = p[10]; t.ecWidth = p[11];
    return t;
}

// k_fa_edges_filter on edge tiles: the end points, the face vertex averages and the cell centres the tile's
// edges need are staged in LDS (the per-edge form gathers ~10 records of 24 bytes per edge from global memory
// and is bound by those gathers).  Ring order as in k_fa_edges: cell i lies between ring faces i and i+1.
template <int T>
__global__ void __launch_bounds__(T) k_fa_filter_tile(State s, Prm prm, EdgeTileView g, const int* edges, uint8_t* faMaybe, int nLaunch, int xcdMap) {
    if (s.acc->stop) return;
    const int li = launchTile(nLaunch, xcdMap);
    if (li < 0) return;
    extern __shared__ double lds[];
    const int tile = li, tid = threadIdx.x;
    // prologue in two dependent round trips (see smoothStage): (1) the three id lists, the edge's id, end-point slots and the
    // first two chunks of its face and cell rows; (2) the records
    const EdgeTileMeta tm = loadTileMeta(g, tile);
    // (laid out by the tile's own counts, see smoothLds: the launch's LDS is 24 B x the largest record total of one tile)
    double* px = lds;                 double* py = px + tm.nPts;     double* pz = py + tm.nPts;
    double* fx = pz + tm.nPts;        double* fy = fx + tm.nFaces;   double* fz = fy + tm.nFaces;
    double* cx = fz + tm.nFaces;      double* cy = cx + tm.nCells;   double* cz = cy + tm.nCells;
    const bool mine = tid < tm.nEdges;
    const int wf4 = tm.efWidth >> 2, wc4 = tm.ecWidth >> 2;
    const ushort4* fRow = reinterpret_cast<const ushort4*>(g.efEll + tm.efBase) + tid;
    const ushort4* cRow = reinterpret_cast<const ushort4*>(g.ecEll + tm.ecBase) + tid;
    const ushort4 padq = make_ushort4(0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF);
    unsigned ep = 0;
    ushort4 f0 = padq, f1 = padq, c0 = padq, c1 = padq;
    if (tm.nFaces <= 3 * T && tm.nCells <= 2 * T && tm.nPts <= 2 * T) {
        const int *idF = g.tfIds + tm.tfOff, *idC = g.tcIds + tm.tcOff, *idP = g.tpIds + tm.tpOff;
        int a[3], b[2], c[2];
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = u * T + tid; a[u] = (i < tm.nFaces) ? idF[i] : -1; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; b[u] = (i < tm.nCells) ? idC[i] : -1; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; c[u] = (i < tm.nPts) ? idP[i] : -1; }
        const int lane = mine ? tid : 0;
        const int ei = tm.edgeBeg + lane;
        const unsigned ep1 = reinterpret_cast<const unsigned*>(g.epLoc)[ei];
        const ushort4* fl = reinterpret_cast<const ushort4*>(g.efEll + tm.efBase) + lane;
        const ushort4* cl = reinterpret_cast<const ushort4*>(g.ecEll + tm.ecBase) + lane;
        const ushort4 g0 = fl[0], g1 = fl[wf4 > 1 ? T : 0], h0 = cl[0], h1 = cl[wc4 > 1 ? T : 0];
        V3 va[3], vb[2], vc[2];
#pragma unroll
        for (int u = 0; u < 3; ++u) va[u] = ldv(s.fAvg, a[u] >= 0 ? a[u] : 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) vb[u] = ldv(s.cellCtr, b[u] >= 0 ? b[u] : 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) vc[u] = ldv(s.ptsCur, c[u] >= 0 ? c[u] : 0);
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = u * T + tid; if (a[u] >= 0) { fx[i] = va[u].x; fy[i] = va[u].y; fz[i] = va[u].z; } }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; if (b[u] >= 0) { cx[i] = vb[u].x; cy[i] = vb[u].y; cz[i] = vb[u].z; } }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; if (c[u] >= 0) { px[i] = vc[u].x; py[i] = vc[u].y; pz[i] = vc[u].z; } }
        if (mine) {
            ep = ep1;
            f0 = g0; c0 = h0;
            if (wf4 > 1) f1 = g1;
            if (wc4 > 1) c1 = h1;
        }
    } else {
        if (mine) {
            const int ei = tm.edgeBeg + tid;
            ep = reinterpret_cast<const unsigned*>(g.epLoc)[ei];
            if (wf4 > 0) f0 = fRow[0];
            if (wf4 > 1) f1 = fRow[T];
            if (wc4 > 0) c0 = cRow[0];
            if (wc4 > 1) c1 = cRow[T];
        }
        stageRecords<T, 3>(s.fAvg, g.tfIds + tm.tfOff, tm.nFaces, fx, fy, fz, tid);
        stageRecords<T, 2>(s.cellCtr, g.tcIds + tm.tcOff, tm.nCells, cx, cy, cz, tid);
        stageRecords<T, 2>(s.ptsCur, g.tpIds + tm.tpOff, tm.nPts, px, py, pz, tid);
    }
    __syncthreads();
    if (!mine) return;
#ifdef SMGPU_FA_FILTER_ABLATE      // (measurement build: staging only)
    if (px[0] == 1.2345e300) markUnsureEdge(s, edges, g.order[tm.edgeBeg + tid], faMaybe);
    return;
#endif
    uint8_t flag = 1;
    if (f0.x != kPad) {
        const V3 e0 = ldsv(px, py, pz, ep & 0xffff), e1 = ldsv(px, py, pz, ep >> 16);
        const V3 cC = 0.5 * (e0 + e1);
        bool ok = true, inside = true;
        const F3 eV = funit(f3(e1 - e0), ok);
        F3 first = {0, 0, 0}, prev = {0, 0, 0}, cPend = {0, 0, 0};
        bool pend = false;
#define SMGPU_FA_PROJECT(OUT, C)                                                        \
        {                                                                               \
            const F3 d_ = f3((C) - cC);                                                 \
            const float t_ = fdot(d_, eV);                                              \
            const F3 w_ = fsubScaled(d_, t_, eV);                                       \
            ok = ok && (fdot(w_, w_) > 1.0e-4f * fdot(d_, d_));                        \
            OUT = funit(w_, ok);                                                        \
        }
#define SMGPU_FA_EVAL(NEXT)                                                             \
        {                                                                               \
            const float a_ = fdot(prev, cPend), b_ = fdot(cPend, (NEXT));               \
            ok = ok && (fabsf(a_) < kNearClamp) && (fabsf(b_) < kNearClamp);            \
            inside = inside && faSumInside(a_, b_, prm);                                \
        }
#define SMGPU_FA_STEP(J, FE, CE)                                                        \
        if ((FE) != kPad) {                                                             \
            F3 nx_;                                                                     \
            SMGPU_FA_PROJECT(nx_, ldsv(fx, fy, fz, (FE)))                               \
            if ((J) == 0) { first = nx_; }                                              \
            else if (pend) { SMGPU_FA_EVAL(nx_) pend = false; }                          \
            prev = nx_;                                                                 \
        }                                                                               \
        if ((CE) != kPad) { SMGPU_FA_PROJECT(cPend, ldsv(cx, cy, cz, (CE))) pend = true; }
#define SMGPU_FA_CHUNK(CIDX, FQ, CQ)                                                    \
        SMGPU_FA_STEP(4 * (CIDX) + 0, (FQ).x, (CQ).x) SMGPU_FA_STEP(4 * (CIDX) + 1, (FQ).y, (CQ).y)   \
        SMGPU_FA_STEP(4 * (CIDX) + 2, (FQ).z, (CQ).z) SMGPU_FA_STEP(4 * (CIDX) + 3, (FQ).w, (CQ).w)
        SMGPU_FA_CHUNK(0, f0, c0)
        if (wf4 > 1) { SMGPU_FA_CHUNK(1, f1, c1) }
        for (int c = 2; c < wf4; ++c) {
            const ushort4 fq = fRow[(size_t)c * T];
            const ushort4 cl_ = cRow[(size_t)(c < wc4 ? c : 0) * T];   // (a select between a global and a private address would become a flat load)
            const ushort4 cq = (c < wc4) ? cl_ : padq;
            SMGPU_FA_CHUNK(c, fq, cq)
        }
        if (pend) { SMGPU_FA_EVAL(first) }   // closed ring: the last cell lies between the last and the first face
#undef SMGPU_FA_CHUNK
#undef SMGPU_FA_STEP
#undef SMGPU_FA_EVAL
#undef SMGPU_FA_PROJECT
        if (ok && inside) flag = 0;
    }
    if (flag) markUnsureEdge(s, edges, g.order[tm.edgeBeg + tid], faMaybe);   // (the edge's id is needed here only: 4 bytes per edge not read)
}

}  // namespace smgpu
