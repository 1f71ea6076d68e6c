// kernels_tiled.hpp -- LDS-staged forms of the two gather stages (see tiles.hpp for the idea).
// Same arithmetic, same summation order, same results as k_face_geom + k_cell_centres and k_smooth in
// kernels.hpp; global memory is only streamed (coalesced id lists and sliced-ELL index tables, 8 bytes
// per lane per load), all indexed access is in LDS.
#pragma once
#include "kernels.hpp"

// -DSMGPU_STAGGER=<n> (experiment): the workgroups of a launch's FIRST round start n x 64 cycles apart per wave slot, so that the
// workgroups sharing a CU are not all staging (or all computing) at the same time
#ifndef SMGPU_STAGGER
#define SMGPU_STAGGER 0
#endif
__device__ __forceinline__ void staggerFirstRound(int bid, int resident) {
#if SMGPU_STAGGER
    if (bid < resident) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;      // HW_REG_HW_ID.wave_id
        for (unsigned k = 0; k < slot; ++k) __builtin_amdgcn_s_sleep(SMGPU_STAGGER);
    }
#else
    (void)bid; (void)resident;
#endif
}
// -DSMGPU_ABLATE=<bits>: timing experiments on smoothPoint (WRONG results; never in the product build) -- 1: no square root per
// neighbour, 2: every LDS gather reads record 0 (no bank conflicts), 4: no freeze loop, 8: no cell loop, 16: nothing after the staging;
// k_geom_tile -- 32: nothing after the staging, 64: no cell phase
#ifndef SMGPU_ABLATE
#define SMGPU_ABLATE 0
#endif
namespace smgpu {

constexpr unsigned kPad = 0xFFFFu;

struct GeomTileView {
    const int* cellOrder; const int* cellBeg; const int* tpOff; const int* tpIds; const int* tfOff; const int* tfIds;
    const int* fvBase; const uint8_t* fvWidth; const uint16_t* faceVerts;
    const int* cfBase; const uint8_t* cfWidth; const uint16_t* cellFaces;
    const uint8_t* tileFlags;    // bit0 all faces quadrilaterals, bit1 all cells six-faced: unrolled paths (same arithmetic)
    const int* meta;             // the per-tile scalars of the arrays above in one record per tile (GeomTileMeta)
    int maxPoints, maxFaces;
};
// One tile's scalars, fetched with scalar loads through the constant address space (the tables are written by the host
// before the first launch).  Read from the separate arrays with ordinary loads -- the compiler cannot prove them
// invariant -- they were a chain of dependent vector-memory round trips in front of the staging.
struct GeomTileMeta { int tpOff, nPts, tfOff, nFaces, fvBase, fvWidth, cellBeg, nCells, cfBase, cfWidth, flags, pad; };
constexpr int kGeomMetaInts = 12;
typedef const __attribute__((address_space(4))) int* const_int_ptr;
__device__ __forceinline__ GeomTileMeta loadTileMeta(const GeomTileView& g, int tile) {
    const_int_ptr p = (const_int_ptr)(g.meta + (size_t)kGeomMetaInts * tile);
    GeomTileMeta t;
    t.tpOff = p[0]; t.nPts = p[1]; t.tfOff = p[2]; t.nFaces = p[3]; t.fvBase = p[4]; t.fvWidth = p[5];
    t.cellBeg = p[6]; t.nCells = p[7]; t.cfBase = p[8]; t.cfWidth = p[9]; t.flags = p[10]; t.pad = 0;
    return t;
}

struct SmoothTileView {
    const int* ptOrder; const int* ptBeg; const int* tcOff; const int* tcIds; const int* tnOff; const int* tnIds;
    const uint16_t* selfLoc;
    const int* pcBase; const uint8_t* pcWidth; const uint16_t* pcEll;
    const int* ppBase; const uint8_t* ppWidth; const uint16_t* ppEll;   // bit 15: the neighbour is an internal point
    const uint16_t* pairEll;
    const int* pfBase; const uint8_t* pfWidth; const uint16_t* pfEll;   // (prev, next) vertex per (point, face)
    const int* meta;             // the per-tile scalars of the arrays above, one record per tile (SmoothTileMeta)
    int maxCells, maxPoints, usePairShare;
};

// a ? x : y member by member (workgroup-uniform a: scalar selects).  `a ? x : y` on the two OBJECTS is a select of their addresses,
// for which the compiler copies both kernel arguments to scratch memory (160 bytes per lane in k_smooth_halo)
__device__ __forceinline__ SmoothTileView pickView(bool a, const SmoothTileView& x, const SmoothTileView& y) {
    SmoothTileView v;
#define SMGPU_PICK(F) v.F = a ? x.F : y.F;
    SMGPU_PICK(ptOrder) SMGPU_PICK(ptBeg) SMGPU_PICK(tcOff) SMGPU_PICK(tcIds) SMGPU_PICK(tnOff) SMGPU_PICK(tnIds) SMGPU_PICK(selfLoc)
    SMGPU_PICK(pcBase) SMGPU_PICK(pcWidth) SMGPU_PICK(pcEll) SMGPU_PICK(ppBase) SMGPU_PICK(ppWidth) SMGPU_PICK(ppEll) SMGPU_PICK(pairEll)
    SMGPU_PICK(pfBase) SMGPU_PICK(pfWidth) SMGPU_PICK(pfEll) SMGPU_PICK(meta) SMGPU_PICK(maxCells) SMGPU_PICK(maxPoints) SMGPU_PICK(usePairShare)
#undef SMGPU_PICK
    return v;
}

struct SmoothTileMeta { int ptBeg, nPts, tcOff, nCells, tnOff, nNbrs, pcBase, pcWidth, ppBase, ppWidth, pfBase, pfWidth; };
constexpr int kSmoothMetaInts = 12;
__device__ __forceinline__ SmoothTileMeta loadTileMeta(const SmoothTileView& g, int tile) {
    const_int_ptr p = (const_int_ptr)(g.meta + (size_t)kSmoothMetaInts * tile);
    SmoothTileMeta t;
    t.ptBeg = p[0]; t.nPts = p[1]; t.tcOff = p[2]; t.nCells = p[3]; t.tnOff = p[4]; t.nNbrs = p[5];
    t.pcBase = p[6]; t.pcWidth = p[7]; t.ppBase = p[8]; t.ppWidth = p[9]; t.pfBase = p[10]; t.pfWidth = p[11];
    return t;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  With xcdMap the launch has 8*ceil(n/8)
// workgroups and XCD x walks the contiguous range [x*ceil(n/8), ...) of the Morton-ordered tile sequence, so
// that tiles which stage the same records share an L2.  Returns -1 for the padding workgroups.
__device__ __forceinline__ int launchTile(int n, int xcdMap, int b = (int)blockIdx.x) {
    if (!xcdMap) return b;
    const int per = (n + 7) >> 3;
    const int i = (b & 7) * per + (b >> 3);
    return i < n ? i : -1;
}
__host__ __device__ inline int tileGrid(int n, int xcdMap) { return xcdMap ? ((n + 7) >> 3) << 3 : n; }

__device__ __forceinline__ V3 ldsv(const double* x, const double* y, const double* z, int i) { return v3(x[i], y[i], z[i]); }

// four square roots behind one range test
__device__ __forceinline__ void sqrtExact4(double m0, double m1, double m2, double m3, double& a0, double& a1, double& a2, double& a3) {
    const unsigned h0 = (unsigned)__double2hiint(m0), h1 = (unsigned)__double2hiint(m1), h2 = (unsigned)__double2hiint(m2),
                   h3 = (unsigned)__double2hiint(m3);
    const unsigned lo = min(min(h0, h1), min(h2, h3)), hi = max(max(h0, h1), max(h2, h3));
    if (SMGPU_FPEXACT_FAST && __builtin_expect(lo >= ((1023u - 767u) << 20) && hi < (2047u << 20), 1)) {
        a0 = sqrtCore(m0); a1 = sqrtCore(m1); a2 = sqrtCore(m2); a3 = sqrtCore(m3);
    } else {
        a0 = sqrt(m0); a1 = sqrt(m1); a2 = sqrt(m2); a3 = sqrt(m3);
    }
}

// Visit the entries of one ELL row (4 per 8-byte chunk, `w4` chunks `stride` chunks apart) in list
// order; pads (0xFFFF) only occur at the tail and are skipped.  No early exit, so the chunk loads
// stay independent of the entry processing.  BODY sees `j` (list position) and `e` (entry value).
// (A macro, not a lambda: closures captured by reference ended up in scratch memory.)
#define SMGPU_ELL_ENTRY(J, E, BODY) { const int j = (J); const unsigned e = (E); if (e != kPad) { BODY } }
#define SMGPU_ELL_FOREACH(ROWS, W4, STRIDE, BODY)                         \
    for (int c_ = 0; c_ < (W4); ++c_) {                                   \
        const ushort4 q_ = (ROWS)[(size_t)c_ * (STRIDE)];                 \
        SMGPU_ELL_ENTRY(4 * c_ + 0, q_.x, BODY)                           \
        SMGPU_ELL_ENTRY(4 * c_ + 1, q_.y, BODY)                           \
        SMGPU_ELL_ENTRY(4 * c_ + 2, q_.z, BODY)                           \
        SMGPU_ELL_ENTRY(4 * c_ + 3, q_.w, BODY)                           \
    }

// Same, with the first two chunks already in registers (Q0, Q1: loaded up front so that their latency
// overlaps the LDS staging); rows wider than 8 entries fetch the remaining chunks here.
#define SMGPU_ELL_CHUNK(C, Q, BODY)                                       \
    {                                                                     \
        SMGPU_ELL_ENTRY(4 * (C) + 0, (Q).x, BODY)                         \
        SMGPU_ELL_ENTRY(4 * (C) + 1, (Q).y, BODY)                         \
        SMGPU_ELL_ENTRY(4 * (C) + 2, (Q).z, BODY)                         \
        SMGPU_ELL_ENTRY(4 * (C) + 3, (Q).w, BODY)                         \
    }
#define SMGPU_ELL_FOREACH_PRE(Q0, Q1, ROWS, W4, STRIDE, BODY)             \
    {                                                                     \
        if ((W4) > 0) SMGPU_ELL_CHUNK(0, Q0, BODY)                        \
        if ((W4) > 1) SMGPU_ELL_CHUNK(1, Q1, BODY)                        \
        for (int c_ = 2; c_ < (W4); ++c_) {                               \
            const ushort4 q_ = (ROWS)[(size_t)c_ * (STRIDE)];             \
            SMGPU_ELL_CHUNK(c_, q_, BODY)                                 \
        }                                                                 \
    }

// Stage n elements (24-byte records picked by an ascending id list) into SoA LDS arrays.  All id loads
// of a thread are issued first, then all record loads, then the LDS stores: two memory round trips per
// tile instead of two per 256 elements.
template <int T, int ROUNDS, int STRIDE = 1>
__device__ __forceinline__ void stageRecords(const double* __restrict__ src, const int* __restrict__ ids, int n, double* x,
                                             double* y, double* z, int tid) {
    for (int base = 0; base < n; base += T * ROUNDS) {
        int id[ROUNDS];
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int i = base + u * T + tid;
            id[u] = (i < n) ? ids[i] : -1;
        }
        V3 v[ROUNDS];
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) v[u] = (id[u] >= 0) ? ldv(src, id[u]) : v3(0, 0, 0);
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int i = base + u * T + tid;
            if (id[u] >= 0) { x[STRIDE * i] = v[u].x; y[STRIDE * i] = v[u].y; z[STRIDE * i] = v[u].z; }
        }
    }
}

// Two (or three) record sets in one pass: all id loads, then all record loads, then the LDS stores -- two
// memory round trips per tile in total.  Falls back to set-by-set staging when a set exceeds its rounds.
template <int T, int R1, int R2>
__device__ __forceinline__ void stageRecords2(const double* __restrict__ s1, const int* __restrict__ i1, int n1, double* x1, double* y1,
                                              double* z1, const double* __restrict__ s2, const int* __restrict__ i2, int n2, double* x2,
                                              double* y2, double* z2, int tid) {
    if (n1 > T * R1 || n2 > T * R2) {
        stageRecords<T, R1>(s1, i1, n1, x1, y1, z1, tid);
        stageRecords<T, R2>(s2, i2, n2, x2, y2, z2, tid);
        return;
    }
    int a[R1], b[R2];
#pragma unroll
    for (int u = 0; u < R1; ++u) { const int i = u * T + tid; a[u] = (i < n1) ? i1[i] : -1; }
#pragma unroll
    for (int u = 0; u < R2; ++u) { const int i = u * T + tid; b[u] = (i < n2) ? i2[i] : -1; }
    V3 va[R1], vb[R2];
#pragma unroll
    for (int u = 0; u < R1; ++u) va[u] = (a[u] >= 0) ? ldv(s1, a[u]) : v3(0, 0, 0);
#pragma unroll
    for (int u = 0; u < R2; ++u) vb[u] = (b[u] >= 0) ? ldv(s2, b[u]) : v3(0, 0, 0);
#pragma unroll
    for (int u = 0; u < R1; ++u) { const int i = u * T + tid; if (a[u] >= 0) { x1[i] = va[u].x; y1[i] = va[u].y; z1[i] = va[u].z; } }
#pragma unroll
    for (int u = 0; u < R2; ++u) { const int i = u * T + tid; if (b[u] >= 0) { x2[i] = vb[u].x; y2[i] = vb[u].y; z2[i] = vb[u].z; } }
}

// the same id list into two destinations from two sources (current and proposed coordinates)
template <int T, int R>
__device__ __forceinline__ void stageRecordsPair(const double* __restrict__ s1, const double* __restrict__ s2, const int* __restrict__ ids,
                                                 int n, double* x1, double* y1, double* z1, double* x2, double* y2, double* z2, int tid) {
    for (int base = 0; base < n; base += T * R) {
        int id[R];
#pragma unroll
        for (int u = 0; u < R; ++u) { const int i = base + u * T + tid; id[u] = (i < n) ? ids[i] : -1; }
        V3 va[R], vb[R];
#pragma unroll
        for (int u = 0; u < R; ++u) { va[u] = (id[u] >= 0) ? ldv(s1, id[u]) : v3(0, 0, 0); vb[u] = (id[u] >= 0) ? ldv(s2, id[u]) : v3(0, 0, 0); }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = base + u * T + tid;
            if (id[u] >= 0) { x1[i] = va[u].x; y1[i] = va[u].y; z1[i] = va[u].z; x2[i] = vb[u].x; y2[i] = vb[u].y; z2[i] = vb[u].z; }
        }
    }
}

// records written by other workgroups of the same launch (tiles beyond the fixed staging rounds: rare, plain loop)
template <int T>
__device__ __forceinline__ void stageRecordsCoh(const double* src, const int* __restrict__ ids, int n, double* x, double* y, double* z, int tid) {
    for (int i = tid; i < n; i += T) { const V3 v = ldvCoh(src, ids[i]); x[i] = v.x; y[i] = v.y; z[i] = v.z; }
}

// Cell centres of the current coordinates for one tile of consecutive cells:
// OpenFOAM makeFaceCentresAndAreas + makeCellCentresAndVols (.com v2412), staged through LDS.
// tileList (may be NULL) selects the tiles of this launch (multi-rank: the tiles away from the shared points
// are recomputed ahead, while exchange F is in flight).
// LDS arrays of one geometry tile
// Records, not component arrays: a point is 3 consecutive doubles, a face 6 (centre, area vector), so that one address
// (index * stride, one v_mad) serves all components through the instruction's offset field -- with nine component arrays
// every LDS access carried its own shift-and-add (the kernel is bound by vector instruction issue).  The
// nine pointers stay (x[kGP * i] is component x of point i), set one element apart.  SMGPU_GEOM_AOS=0: component arrays.
#ifndef SMGPU_GEOM_AOS
#define SMGPU_GEOM_AOS 1
#endif
constexpr int kGP = SMGPU_GEOM_AOS ? 3 : 1;   // index stride of a point in px / py / pz
// How the geometry kernel reads a 24-byte record from LDS.  The compiler pairs the x and y reads of a record into one
// ds_read2_b64: 8 LDS cycles per wave at the 32-bank modulus, where three plain ds_read_b64 cost 2 cycles each at the 64-bank
// modulus (MI355X_MICROARCH.md, LDS table).  SMGPU_GEOM_LDS_B64=1 (default) keeps the three reads apart (a volatile access is
// never merged; the waits stay the compiler's): k_geom_tile 50.4 -> 49.1 us on 100^3, unchanged on the 10 M-cell cavity mesh --
// the LDS pipeline is not what the kernel waits for.
#ifndef SMGPU_GEOM_LDS_B64
#define SMGPU_GEOM_LDS_B64 1
#endif
__device__ __forceinline__ V3 ldsg(const double* x, const double* y, const double* z, int i) {
#if SMGPU_GEOM_LDS_B64
    typedef const volatile __attribute__((address_space(3))) double* LdsPtr;   // (a volatile access through a generic pointer would become a flat load)
    const LdsPtr vx = (LdsPtr)x, vy = (LdsPtr)y, vz = (LdsPtr)z;
    return v3(vx[i], vy[i], vz[i]);
#else
    return v3(x[i], y[i], z[i]);
#endif
}
// (a face record is padded to an odd number of doubles: with 6 the records of consecutive faces -- what a wave writes in the
// face phase -- fall on 8 of the 16 bank pairs, a two-way conflict on every access; SMGPU_GEOM_FACE_STRIDE=6 restores that)
// Round 5: back to 6.  With five workgroups per CU (SMGPU_GEOM_WAVES) the LDS footprint decides whether a 128-cell tile fits the
// 31.8 KB line as it is (6: 27.7 KB) or has to be cut earlier (7: 6 % more tiles); same box, alternating three times, 5 workgroups
// both ways: 49.6 -> 47.9 us on 100^3, 538 -> 527 us on the 10 M-cell polyhedral mesh.
#ifndef SMGPU_GEOM_FACE_STRIDE
#define SMGPU_GEOM_FACE_STRIDE 6
#endif
constexpr int kGF = SMGPU_GEOM_AOS ? SMGPU_GEOM_FACE_STRIDE : 1;   // index stride of a face in fcx .. faz
struct GeomLds {
    double *px, *py, *pz;        // the tile's points
    double *fcx, *fcy, *fcz;     // face centres
    double *fax, *fay, *faz;     // face area vectors
};
// (laid out by the TILE's own counts, as smoothLds: the launch's LDS is the largest 3 nPts + kGF nFaces doubles of one tile, which
// GeomTiles::buildBoundaries caps -- capWeighted -- so that FIVE workgroups share a CU)
__device__ __forceinline__ GeomLds geomLds(double* lds, const GeomTileMeta& tm) {
    GeomLds L;
    if (SMGPU_GEOM_AOS) {
        L.px = lds;                      L.py = L.px + 1;  L.pz = L.px + 2;
        L.fcx = L.px + 3 * tm.nPts;      L.fcy = L.fcx + 1; L.fcz = L.fcx + 2;
        L.fax = L.fcx + 3;               L.fay = L.fcx + 4; L.faz = L.fcx + 5;
        return L;
    }
    L.px = lds;                  L.py = L.px + tm.nPts;    L.pz = L.py + tm.nPts;
    L.fcx = L.pz + tm.nPts;      L.fcy = L.fcx + tm.nFaces; L.fcz = L.fcy + tm.nFaces;
    L.fax = L.fcz + tm.nFaces;   L.fay = L.fax + tm.nFaces; L.faz = L.fay + tm.nFaces;
    return L;
}

// face i of the tile: OpenFOAM makeFaceCentresAndAreas on the staged points -> LDS (+ the per-face values the owner's
// tile publishes).  tflags bit0: every face of the tile is a quadrilateral.
// ORG = false: OpenFOAM.com v2312-v2506 (fan triangles weighted by |n|); ORG = true: OpenFOAM.org 12 (fan triangles weighted
// by n . nHat with nHat the normalised sum of the n, centre = point average when the weights sum to <= vSmall) -- the two
// OpenFOAM lines the reference builds against (Allwmake:47); the area vector 0.5 * sum(n) is the same in both.
// pre / preQ: the face's vertex row was read by the kernel's prologue (quadrilateral tiles, first two rounds)
template <bool ORG>
__device__ __forceinline__ void geomFace(const State& s, const GeomTileView& g, const GeomLds& L, const GeomTileMeta& tm, int i, unsigned tflags,
                                         int wantAvg, int writeFaces, bool pre = false, ushort4 preQ = make_ushort4(0, 0, 0, 0)) {
    const double *px = L.px, *py = L.py, *pz = L.pz;
    const int b = tm.tfOff;
    const int fw4 = tm.fvWidth >> 2;
    const ushort4* fvTile = reinterpret_cast<const ushort4*>(g.faceVerts + tm.fvBase);
    V3 fCentre, ctr, area;
    if (!ORG && (tflags & 1u)) {
        // the general loop below unrolled for four vertices -- same operations in the same order, every vertex read
        // once, no pad / position tests
        const ushort4 q = pre ? preQ : fvTile[i];
        const V3 p0 = ldsg(px, py, pz, kGP * (q.x)), p1 = ldsg(px, py, pz, kGP * (q.y)), p2 = ldsg(px, py, pz, kGP * (q.z)), p3 = ldsg(px, py, pz, kGP * (q.w));
        fCentre = divByCount(((p0 + p1) + p2) + p3, 4);
        V3 sumN = v3(0, 0, 0), sumAc = v3(0, 0, 0);
        double sumA = 0.0;
        // the four fan triangles: normals first, their four lengths behind one range test (sqrtExact4), then the sums in
        // the order of the general loop
        const V3 n0 = cross(p1 - p0, fCentre - p0), n1 = cross(p2 - p1, fCentre - p1), n2 = cross(p3 - p2, fCentre - p2),
                 n3 = cross(p0 - p3, fCentre - p3);
        double a0, a1, a2, a3;
        sqrtExact4(magSqr(n0), magSqr(n1), magSqr(n2), magSqr(n3), a0, a1, a2, a3);
#define SMGPU_FAN4(THIS, NEXT, NN, A)                                          \
    {                                                                          \
        const V3 c = ((THIS) + (NEXT)) + fCentre;                              \
        sumN = sumN + (NN);                                                    \
        sumA += (A);                                                           \
        sumAc = sumAc + (A) * c;                                               \
    }
        SMGPU_FAN4(p0, p1, n0, a0) SMGPU_FAN4(p1, p2, n1, a1) SMGPU_FAN4(p2, p3, n2, a2) SMGPU_FAN4(p3, p0, n3, a3)
#undef SMGPU_FAN4
        if (sumA < SMGPU_ROOTVSMALL) { ctr = fCentre; area = v3(0, 0, 0); }
        else { ctr = divExact((1.0 / 3.0) * sumAc, sumA); area = 0.5 * sumN; }
    } else {
        const ushort4* row = fvTile + (size_t)i * fw4;
        // vertex average (fCentre of makeFaceCentresAndAreas; calcFaceCenter SM.C:1103-1130)
        fCentre = v3(0, 0, 0);
        int n = 0;
        SMGPU_ELL_FOREACH(row, fw4, 1, {
            const V3 p = ldsg(px, py, pz, kGP * (e));
            fCentre = (j == 0) ? p : fCentre + p;
            n = j + 1;
        })
        fCentre = divByCount(fCentre, n);
        if (n == 3) {
            const ushort4 q = row[0];
            const V3 p0 = ldsg(px, py, pz, kGP * (q.x)), p1 = ldsg(px, py, pz, kGP * (q.y)), p2 = ldsg(px, py, pz, kGP * (q.z));
            ctr = (1.0 / 3.0) * ((p0 + p1) + p2);
            area = 0.5 * cross(p1 - p0, p2 - p0);
        } else if (ORG) {
            V3 sumA = v3(0, 0, 0);
            V3 first = v3(0, 0, 0), thisPoint = v3(0, 0, 0);
            SMGPU_ELL_FOREACH(row, fw4, 1, {
                const V3 p = ldsg(px, py, pz, kGP * (e));
                if (j == 0) { first = p; thisPoint = p; }
                else { sumA = sumA + cross(p - thisPoint, fCentre - thisPoint); thisPoint = p; }
            })
            sumA = sumA + cross(first - thisPoint, fCentre - thisPoint);
            const V3 sumAHat = sumA / mag(sumA);   // normalised(sumA)
            double sumAn = 0.0;
            V3 sumAnc = v3(0, 0, 0);
#define SMGPU_FAN_ORG(NEXT)                                                    \
    {                                                                          \
        const V3 nextPoint = (NEXT);                                           \
        const V3 a = cross(nextPoint - thisPoint, fCentre - thisPoint);        \
        const V3 c = (thisPoint + nextPoint) + fCentre;                        \
        const double an = dot(a, sumAHat);                                     \
        sumAn += an;                                                           \
        sumAnc = sumAnc + an * c;                                              \
        thisPoint = nextPoint;                                                 \
    }
            SMGPU_ELL_FOREACH(row, fw4, 1, {
                const V3 p = ldsg(px, py, pz, kGP * (e));
                if (j == 0) { first = p; thisPoint = p; }
                else SMGPU_FAN_ORG(p)
            })
            SMGPU_FAN_ORG(first)
#undef SMGPU_FAN_ORG
            if (sumAn > SMGPU_VSMALL) ctr = ((1.0 / 3.0) * sumAnc) / sumAn;
            else ctr = fCentre;
            area = 0.5 * sumA;
        } else {
            V3 sumN = v3(0, 0, 0), sumAc = v3(0, 0, 0);
            double sumA = 0.0;
            V3 first = v3(0, 0, 0), thisPoint = v3(0, 0, 0);
#define SMGPU_FAN(NEXT)                                                        \
    {                                                                          \
        const V3 nextPoint = (NEXT);                                           \
        const V3 c = (thisPoint + nextPoint) + fCentre;                        \
        const V3 nn = cross(nextPoint - thisPoint, fCentre - thisPoint);       \
        const double a = sqrtExact(magSqr(nn));                                \
        sumN = sumN + nn;                                                      \
        sumA += a;                                                             \
        sumAc = sumAc + a * c;                                                 \
        thisPoint = nextPoint;                                                 \
    }
            SMGPU_ELL_FOREACH(row, fw4, 1, {
                const V3 p = ldsg(px, py, pz, kGP * (e));
                if (j == 0) { first = p; thisPoint = p; }
                else SMGPU_FAN(p)
            })
            SMGPU_FAN(first)
#undef SMGPU_FAN
            if (sumA < SMGPU_ROOTVSMALL) { ctr = fCentre; area = v3(0, 0, 0); }
            else { ctr = divExact((1.0 / 3.0) * sumAc, sumA); area = 0.5 * sumN; }
        }
    }
    L.fcx[kGF * i] = ctr.x; L.fcy[kGF * i] = ctr.y; L.fcz[kGF * i] = ctr.z;
    L.fax[kGF * i] = area.x; L.fay[kGF * i] = area.y; L.faz[kGF * i] = area.z;
    if (wantAvg && s.avgPacked) stv(s.fAvg, b + i, fCentre);   // tile order: contiguous stores (the face-angle filter's
                                                                // tiles hold positions into this order)
    if ((wantAvg && !s.avgPacked) || writeFaces) {
        const int fid = g.tfIds[b + i];
        if (fid < 0) {   // this tile holds the face's owner cell: it publishes the per-face values
            const int f = fid & 0x7fffffff;
            if (wantAvg && !s.avgPacked) stv(s.fAvg, f, fCentre);
            if (writeFaces) { stv(s.fCtr, f, ctr); stv(s.fArea, f, area); }
        }
    }
}

// the thread's cell of the tile: OpenFOAM makeCellCentresAndVols (.com v2412) on the face values in LDS.
// tflags bit1: every cell of the tile has six faces.
// what a cell thread reads from global memory: its cell id and (tiles of six-faced cells) its face row; read by the kernel's
// prologue so that the latency runs beside the staging of the points instead of in front of the cell phase
struct GeomCellIn { bool mine; int c; ushort4 qa, qb; };
template <int T>
__device__ __forceinline__ GeomCellIn geomCellLoad(const GeomTileView& g, const GeomTileMeta& tm, int tid, unsigned tflags) {
    GeomCellIn in;
    const int ci = tm.cellBeg + tid;
    in.mine = tid < tm.nCells;
    in.c = 0;
    in.qa = in.qb = make_ushort4(0, 0, 0, 0);
    if (in.mine) {
        in.c = g.cellOrder[ci];
        if ((tflags & 2u)) {
            const ushort4* row = reinterpret_cast<const ushort4*>(g.cellFaces + tm.cfBase) + tid;
            in.qa = row[0]; in.qb = row[T];
        }
    }
    return in;
}
// coh (workgroup-uniform): the cell centre is read by another workgroup of the SAME launch (the pack role of k_geom_halo):
// coherent store.  A run-time flag, not a template parameter: k_geom_halo must hold ONE copy of this code (41 KB; with three
// inlined copies the kernel was 135 KB against a 64 KB instruction cache, and roles running side by side on a CU evicted each
// other's code: 92 us for geometry + pack instead of 51 + 23)
template <int T, bool ORG>
__device__ __forceinline__ void geomCell(const State& s, const GeomTileView& g, const GeomLds& L, const GeomTileMeta& tm, int tid, unsigned tflags, const GeomCellIn& in,
                                         bool coh = false) {
    const double *fcx = L.fcx, *fcy = L.fcy, *fcz = L.fcz, *fax = L.fax, *fay = L.fay, *faz = L.faz;
    if (!in.mine) return;
    const int c = in.c;
    const int cw4 = tm.cfWidth >> 2;
    const ushort4* row = reinterpret_cast<const ushort4*>(g.cellFaces + tm.cfBase) + tid;
    V3 cEst = v3(0, 0, 0), ctr = v3(0, 0, 0);
    double vol = 0.0;
#define SMGPU_PYR(E, FC)                                                                                   \
    {                                                                                                      \
        const V3 fA = ldsg(fax, fay, faz, kGF * ((E) & 0x7fff));                                                   \
        /* neighbour side: Sf . (cEst - Cf) = -(Sf . (Cf - cEst)) bit for bit (IEEE subtraction, multiplication and     \
           addition are odd functions under round-to-nearest) -- one expression and a sign, no divergent branch per face */ \
        double pyr3Vol = dot(fA, (FC) - cEst);                                                             \
        pyr3Vol = ((E) & 0x8000) ? -pyr3Vol : pyr3Vol;                                                     \
        if (ORG) pyr3Vol = (pyr3Vol > SMGPU_VSMALL) ? pyr3Vol : SMGPU_VSMALL;   /* OpenFOAM.org: max(.., vSmall) */ \
        const V3 pc = (3.0 / 4.0) * (FC) + (1.0 / 4.0) * cEst;                                             \
        ctr = ctr + pyr3Vol * pc;                                                                          \
        vol += pyr3Vol;                                                                                    \
    }
    if ((tflags & 2u)) {
        // the loops below unrolled for six faces, each face centre read once
        const ushort4 qa = in.qa, qb = in.qb;
        const unsigned e0 = qa.x, e1 = qa.y, e2 = qa.z, e3 = qa.w, e4 = qb.x, e5 = qb.y;
        const V3 c0 = ldsg(fcx, fcy, fcz, kGF * (e0 & 0x7fff)), c1 = ldsg(fcx, fcy, fcz, kGF * (e1 & 0x7fff)), c2 = ldsg(fcx, fcy, fcz, kGF * (e2 & 0x7fff)),
                 c3 = ldsg(fcx, fcy, fcz, kGF * (e3 & 0x7fff)), c4 = ldsg(fcx, fcy, fcz, kGF * (e4 & 0x7fff)), c5 = ldsg(fcx, fcy, fcz, kGF * (e5 & 0x7fff));
        cEst = cEst + c0; cEst = cEst + c1; cEst = cEst + c2; cEst = cEst + c3; cEst = cEst + c4; cEst = cEst + c5;
        cEst = divExact(cEst, 6.0);
        SMGPU_PYR(e0, c0) SMGPU_PYR(e1, c1) SMGPU_PYR(e2, c2) SMGPU_PYR(e3, c3) SMGPU_PYR(e4, c4) SMGPU_PYR(e5, c5)
    } else {
        int nFaces = 0;
        SMGPU_ELL_FOREACH(row, cw4, T, {
            cEst = cEst + ldsg(fcx, fcy, fcz, kGF * (e & 0x7fff));
            nFaces = j + 1;
        })
        cEst = divByCount(cEst, nFaces);
        SMGPU_ELL_FOREACH(row, cw4, T, {
            (void)j;
            const V3 fc = ldsg(fcx, fcy, fcz, kGF * (e & 0x7fff));
            SMGPU_PYR(e, fc)
        })
    }
#undef SMGPU_PYR
    if (fabs(vol) > SMGPU_VSMALL) ctr = divExact(ctr, vol);
    else ctr = cEst;
    if (coh) stvCoh(s.cellCtr, c, ctr); else stv(s.cellCtr, c, ctr);
}

// deferN > 0: the first workgroup also closes the PREVIOUS iteration (reduction of its deferN workgroup partials into
// stats[deferIter], reset of the per-iteration counters) -- with relTol <= 0 nothing can stop the loop, so that work does
// not need a launch of its own between the iterations (k_finish, ~6 us of launch latency per iteration).
// (waves per SIMD the geometry kernels are compiled for.  Round 5: 5 -- 96 VGPRs instead of 118, no spills -- together with an LDS
// footprint of at most 31.8 KB per tile: same box, 555 -> 538 us on the 10 M-cell polyhedral mesh, nothing on the 1 M-cell block)
#ifndef SMGPU_GEOM_WAVES
#define SMGPU_GEOM_WAVES 5
#endif
// ---- the workgroup's work ---------------------------------------------------------------------------------------------------
// round 1 of the prologue for one tile: the point id list (first two rounds of T)
template <int T>
__device__ __forceinline__ void geomLoadIds(const GeomTileView& g, const GeomTileMeta& tm, int tid, int (&id)[2]) {
    const int* ids = g.tpIds + tm.tpOff;
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int i = u * T + tid; id[u] = (i < tm.nPts) ? ids[i] : -1; }
}
// ... and what the face and cell phases read from global memory per thread: the vertex rows of the first two face rounds of a
// quadrilateral tile, the cell's id and face row
struct GeomRows { ushort4 fq[2]; GeomCellIn cin; bool preFaces; };
template <int T, bool ORG>
__device__ __forceinline__ GeomRows geomLoadRows(const GeomTileView& g, const GeomTileMeta& tm, int tid) {
    GeomRows r;
    const unsigned tflags = (unsigned)tm.flags;
    r.preFaces = !ORG && (tflags & 1u);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = u * T + tid;
        r.fq[u] = (r.preFaces && i < tm.nFaces) ? reinterpret_cast<const ushort4*>(g.faceVerts + tm.fvBase)[i] : make_ushort4(0, 0, 0, 0);
    }
    r.cin = geomCellLoad<T>(g, tm, tid, tflags);
    return r;
}
// round 2: the point records of the ids -> registers
__device__ __forceinline__ void geomLoadPoints(const State& s, const int (&id)[2], V3 (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = (id[u] >= 0) ? ldv(s.ptsCur, id[u]) : v3(0, 0, 0);
}
template <int T>
__device__ __forceinline__ void geomStorePoints(const State& s, const GeomTileView& g, const GeomTileMeta& tm, const GeomLds& L, const int (&id)[2],
                                                const V3 (&v)[2], int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = u * T + tid;
        if (id[u] >= 0) { L.px[kGP * i] = v[u].x; L.py[kGP * i] = v[u].y; L.pz[kGP * i] = v[u].z; }
    }
    if (tm.nPts > 2 * T)      // tiles beyond the fixed number of rounds
        stageRecords<T, 2, kGP>(s.ptsCur, g.tpIds + tm.tpOff + 2 * T, tm.nPts - 2 * T, L.px + kGP * 2 * T, L.py + kGP * 2 * T, L.pz + kGP * 2 * T, tid);
}
// phase 1: every face of the tile once -> LDS
template <int T, bool ORG>
__device__ __forceinline__ void geomFaces(const State& s, const GeomTileView& g, const GeomLds& L, const GeomTileMeta& tm, const GeomRows& r, int tid,
                                          int wantAvg, int writeFaces) {
    const unsigned tflags = (unsigned)tm.flags;
    const int nf = tm.nFaces;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = u * T + tid;
        if (i < nf) geomFace<ORG>(s, g, L, tm, i, tflags, wantAvg, writeFaces, r.preFaces, r.fq[u]);
    }
    for (int i = 2 * T + tid; i < nf; i += T) geomFace<ORG>(s, g, L, tm, i, tflags, wantAvg, writeFaces);
}

// bid = the workgroup's index among the geometry workgroups of the launch.  (Two tiles per workgroup with both prologues in
// flight at once was built and measured: no gain on hex meshes, 17 % slower on the polyhedral one -- DESIGN 9.)
template <int T, bool ORG>
__device__ __forceinline__ void geomTileBody(const MeshView& m, const State& s, const GeomTileView& g, int wantAvg, int writeFaces, const int* tileList,
                                             int nLaunch, int xcdMap, int deferN, int deferIter, double* deferLocal, double* deferHist, int bid, bool coh = false) {
    // the "loop has stopped" flag (relTol reached, SM.C:2401) is tested after the staging: a dependent global load in front of
    // everything else put its latency on every workgroup's critical path; a stopped run stages one tile in vain
    const int stopped = s.acc->stop;
    const int li = launchTile(nLaunch, xcdMap, bid);
    if (li < 0) return;
    staggerFirstRound(bid, 256 * 4);
    if (deferN > 0 && bid == 0) { if (stopped) return; finishPartials<T>(s, deferN, deferIter, -1.0, deferLocal, deferHist); __syncthreads(); }
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int tile = tileList ? ((const_int_ptr)tileList)[li] : li;
    const GeomTileMeta tm = loadTileMeta(g, tile);
    const GeomLds L = geomLds(lds, tm);
    int id[2];
    geomLoadIds<T>(g, tm, tid, id);                               // round 1 (the id list first: the point records depend on it)
    const GeomRows r = geomLoadRows<T, ORG>(g, tm, tid);
    V3 v[2];
    geomLoadPoints(s, id, v);                                     // round 2
    geomStorePoints<T>(s, g, tm, L, id, v, tid);
    if (stopped) return;
    __syncthreads();
    if (SMGPU_ABLATE & 32) { if (tid < tm.nCells) stv(s.cellCtr, ((const_int_ptr)g.cellOrder)[tm.cellBeg + tid], v3(L.px[tid], 0.0, 0.0)); return; }   // (timing experiment: staging only)
    geomFaces<T, ORG>(s, g, L, tm, r, tid, wantAvg, writeFaces);  // phase 1: every face of the tile once
    __syncthreads();
    if (SMGPU_ABLATE & 64) return;                                // (timing experiment: no cell phase)
    geomCell<T, ORG>(s, g, L, tm, tid, (unsigned)tm.flags, r.cin, coh);   // phase 2: one thread per cell
}
template <int T, bool ORG>
__global__ void __launch_bounds__(T, (SMGPU_GEOM_WAVES * T) / 256) k_geom_tile(MeshView m, State s, GeomTileView g, int wantAvg, int writeFaces, const int* tileList,
                                                  int nLaunch, int xcdMap, int deferN, int deferIter, double* deferLocal,
                                                  double* deferHist) {
    geomTileBody<T, ORG>(m, s, g, wantAvg, writeFaces, tileList, nLaunch, xcdMap, deferN, deferIter, deferLocal, deferHist, (int)blockIdx.x);
}

// The fused per-point proposal kernel of kernels.hpp (k_smooth) with the cell centres and neighbour
// coordinates of one tile of consecutive points staged in LDS.
// tileList (may be NULL) selects the tiles of this launch: the multi-rank driver smooths the tiles without
// shared points while exchange A is in flight, the others after it.
// LDS arrays of one smoothing tile and the per-thread prologue (what a thread reads from global memory besides the
// staged records: point id, own LDS slot, the first two ELL chunks of both rows)
struct SmoothLds { double *cx, *cy, *cz, *nx, *ny, *nz; };
// The arrays are laid out by the TILE's own record counts (wave-uniform, from its meta record), not by the maxima over all tiles:
// the launch's LDS size is then 24 B x the largest (cells + neighbours) of any ONE tile -- which the greedy boundary pass caps
// (SmoothTiles::buildBoundaries, capTotal) -- instead of the sum of two maxima reached by different tiles.
__device__ __forceinline__ SmoothLds smoothLds(double* lds, const SmoothTileMeta& tm) {
    SmoothLds L;
    L.cx = lds;                 L.cy = L.cx + tm.nCells;  L.cz = L.cy + tm.nCells;
    L.nx = L.cz + tm.nCells;    L.ny = L.nx + tm.nNbrs;   L.nz = L.ny + tm.nNbrs;
    return L;
}
// The three shortest incident edges in list order (stable: a later edge of equal length stays behind, SM.C:325-387), kept as
// (length, word) with word = list position << 16 | ELL entry (0xFFFFFFFF: empty).  The reference's if / else-if chain
//     if (k1 < 0 || len < l1) {3 <- 2, 2 <- 1, 1 <- new} else if (k2 < 0 || len < l2) {3 <- 2, 2 <- new} else if (k3 < 0 || len < l3) {3 <- new}
// as selects: b1, b2, b3 are the chain's three branches, so the result is the chain's for every input (NaN lengths included).
// The nested branches cost ~33 register moves per neighbour (the compiler rotates the nine values at every join, and in a wave
// of 64 lanes every level is taken by someone); the selects are 15.
struct Top3 {
    double l1 = 0, l2 = 0, l3 = 0;
    unsigned w1 = 0xFFFFFFFFu, w2 = 0xFFFFFFFFu, w3 = 0xFFFFFFFFu;
    __device__ __forceinline__ void offer(double len, int j, unsigned e, bool take) {
        const unsigned w = ((unsigned)j << 16) | e;
        const bool b1 = take && (w1 == 0xFFFFFFFFu || len < l1);
        const bool b2 = take && !b1 && (w2 == 0xFFFFFFFFu || len < l2);
        const bool b3 = take && !b1 && !b2 && (w3 == 0xFFFFFFFFu || len < l3);
        const bool b12 = b1 || b2;
        l3 = b12 ? l2 : (b3 ? len : l3);  w3 = b12 ? w2 : (b3 ? w : w3);
        l2 = b1 ? l1 : (b2 ? len : l2);   w2 = b1 ? w1 : (b2 ? w : w2);
        l1 = b1 ? len : l1;               w1 = b1 ? w : w1;
    }
    __device__ __forceinline__ bool has2() const { return w2 != 0xFFFFFFFFu; }
    __device__ __forceinline__ bool has3() const { return w3 != 0xFFFFFFFFu; }
    __device__ __forceinline__ int k1() const { return (int)(w1 >> 16); }
    __device__ __forceinline__ int k2() const { return (int)(w2 >> 16); }
};
struct SmoothRow {
    bool mine; int p, selfL, wn4, wc4, slot;
    int peer, dst0, ndst;  // shared-point tiles: the two-sharer point's peer code (State::spPeer) or -1, first send slot, send slots
    unsigned fl;
    const ushort4 *ppRow, *pcRow;
    ushort4 pp0, pp1, pc0, pc1;
    ushort4 pe0, pe1;      // first two chunks of the common-cell bit row (pairEll), when usePairShare
};
// the common-cell bits of neighbour k (one bit per other neighbour), from the prologue's registers for the first 8 neighbours
template <int T>
__device__ __forceinline__ unsigned pairBits(const SmoothTileView& g, const SmoothTileMeta& tm, const SmoothRow& R, int tid, int k) {
    if (k < 8) {
        const ushort4 q = (k & 4) ? R.pe1 : R.pe0;
        const unsigned lo = (k & 1) ? q.y : q.x, hi = (k & 1) ? q.w : q.z;
        return (k & 2) ? hi : lo;
    }
    const uint16_t* pe = g.pairEll + tm.ppBase;
    return pe[((size_t)(k >> 2) * T + tid) * 4 + (k & 3)];
}
// Prologue of the kernels on smoothing tiles: the tile's cell centres and neighbour coordinates into LDS and the thread's own
// inputs into registers, in TWO dependent memory round trips: (1) the id lists of both record sets, the thread's point id,
// LDS slot and first two chunks of both ELL rows; (2) the records, the point's flags and its shared-point slot.  No branch
// between the loads of a round (lanes without a point read the row of lane 0): conditional loads made the compiler wait for
// each of them separately -- eight round trips per tile before.
// COHC: the cell centres were written by other workgroups of the SAME launch (pack role of k_geom_halo): coherent loads.
// SLOTS: where a point's shared-point slot comes from -- 0: nowhere (a tile without shared points), 1: State::sharedSlot[p]
// (a load that depends on the point id: second round), 2: tables by tile position (first round): `tb`, the shared points' own
// tiles (State::spSlot / spPeer / spDst0 / spNDst) or the regular tiles of k_smooth_halo (State::posSlot only)
struct SlotTabs { const int* slot; const int* peer; const int* dst0; const int* ndst; };
template <int T, bool COHC = false, int SLOTS = 1>
__device__ __forceinline__ SmoothRow smoothStage(const MeshView& m, const State& s, const SmoothTileView& g, const SmoothTileMeta& tm,
                                                 const SmoothLds& L, int tid, const SlotTabs& tb = SlotTabs{nullptr, nullptr, nullptr, nullptr}) {
    SmoothRow r;
    r.peer = -1; r.dst0 = -1; r.ndst = 0;
    r.mine = tid < tm.nPts;
    r.wn4 = tm.ppWidth >> 2; r.wc4 = tm.pcWidth >> 2;
    const int lane = r.mine ? tid : 0;
    r.ppRow = reinterpret_cast<const ushort4*>(g.ppEll + tm.ppBase) + tid;
    r.pcRow = reinterpret_cast<const ushort4*>(g.pcEll + tm.pcBase) + tid;
    const bool fast = tm.nCells <= 2 * T && tm.nNbrs <= 3 * T && r.wn4 > 0 && r.wc4 > 0;
    const ushort4 padq = make_ushort4(0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF);
    if (fast) {
        const int* ic = g.tcIds + tm.tcOff;
        const int* in = g.tnIds + tm.tnOff;
        int a[2], b[3];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; a[u] = (i < tm.nCells) ? ic[i] : -1; }
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = u * T + tid; b[u] = (i < tm.nNbrs) ? in[i] : -1; }
        const int pi = tm.ptBeg + lane;
        const int p = g.ptOrder[pi];
        const int selfL = g.selfLoc[pi];
        const ushort4* ppl = reinterpret_cast<const ushort4*>(g.ppEll + tm.ppBase) + lane;
        const ushort4* pcl = reinterpret_cast<const ushort4*>(g.pcEll + tm.pcBase) + lane;
        const ushort4 pp0 = ppl[0], pp1 = ppl[r.wn4 > 1 ? T : 0], pc0 = pcl[0], pc1 = pcl[r.wc4 > 1 ? T : 0];
        const ushort4* pel = reinterpret_cast<const ushort4*>((g.usePairShare ? g.pairEll : g.ppEll) + tm.ppBase) + lane;
        r.pe0 = pel[0]; r.pe1 = pel[r.wn4 > 1 ? T : 0];
        int slot = -1, peer = -1, dst0 = -1, ndst = 0;
        if (SLOTS == 2) { slot = tb.slot[pi]; if (tb.peer) { peer = tb.peer[pi]; dst0 = tb.dst0[pi]; ndst = tb.ndst[pi]; } }
        // round 2
        V3 va[2], vb[3];
#pragma unroll
        for (int u = 0; u < 2; ++u) va[u] = (a[u] >= 0) ? (COHC ? ldvCoh(s.cellCtr, a[u]) : ldv(s.cellCtr, a[u])) : v3(0, 0, 0);
#pragma unroll
        for (int u = 0; u < 3; ++u) vb[u] = (b[u] >= 0) ? ldv(s.ptsCur, b[u]) : v3(0, 0, 0);
        const unsigned fl = m.pflags[p];
        if (SLOTS == 1) slot = s.sharedSlot ? s.sharedSlot[p] : -1;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * T + tid; if (a[u] >= 0) { L.cx[i] = va[u].x; L.cy[i] = va[u].y; L.cz[i] = va[u].z; } }
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = u * T + tid; if (b[u] >= 0) { L.nx[i] = vb[u].x; L.ny[i] = vb[u].y; L.nz[i] = vb[u].z; } }
        r.p = r.mine ? p : 0; r.selfL = r.mine ? selfL : 0;
        r.fl = fl; r.slot = r.mine ? slot : -1; r.peer = r.mine ? peer : -1; r.dst0 = dst0; r.ndst = r.mine ? ndst : 0;
        r.pp0 = r.mine ? pp0 : padq; r.pp1 = (r.mine && r.wn4 > 1) ? pp1 : padq;
        r.pc0 = r.mine ? pc0 : padq; r.pc1 = (r.mine && r.wc4 > 1) ? pc1 : padq;
        return r;
    }
    // tiles beyond the fixed number of staging rounds (or with empty rows): set by set
    r.p = 0; r.selfL = 0; r.fl = 0; r.slot = -1;
    r.pp0 = padq; r.pp1 = padq; r.pc0 = padq; r.pc1 = padq; r.pe0 = padq; r.pe1 = padq;
    if (r.mine) {
        if (g.usePairShare) {
            const ushort4* peRow = reinterpret_cast<const ushort4*>(g.pairEll + tm.ppBase) + tid;
            if (r.wn4 > 0) r.pe0 = peRow[0];
            if (r.wn4 > 1) r.pe1 = peRow[T];
        }
        const int pi = tm.ptBeg + tid;
        r.p = g.ptOrder[pi];
        r.selfL = g.selfLoc[pi];
        if (r.wn4 > 0) r.pp0 = r.ppRow[0];
        if (r.wn4 > 1) r.pp1 = r.ppRow[T];
        if (r.wc4 > 0) r.pc0 = r.pcRow[0];
        if (r.wc4 > 1) r.pc1 = r.pcRow[T];
        r.fl = m.pflags[r.p];
        if (SLOTS == 1) r.slot = s.sharedSlot ? s.sharedSlot[r.p] : -1;
        if (SLOTS == 2) { r.slot = tb.slot[pi]; if (tb.peer) { r.peer = tb.peer[pi]; r.dst0 = tb.dst0[pi]; r.ndst = tb.ndst[pi]; } }
    }
    if (COHC) stageRecordsCoh<T>(s.cellCtr, g.tcIds + tm.tcOff, tm.nCells, L.cx, L.cy, L.cz, tid);
    else stageRecords<T, 2>(s.cellCtr, g.tcIds + tm.tcOff, tm.nCells, L.cx, L.cy, L.cz, tid);
    stageRecords<T, 3>(s.ptsCur, g.tnIds + tm.tnOff, tm.nNbrs, L.nx, L.ny, L.nz, tid);
    return r;
}

// the thread's point of the tile, from the staged cell centres / neighbour coordinates (see k_smooth for the steps)
// MODE 0: a shared point's combined record comes from State::combA (k_halo_combineA*); 1: the tile holds no shared point;
// 2 (k_smooth_halo; ONE copy of the code for both kinds of tiles, see geomCell) -- spTile, the shared points' own tiles: the
// combined record of a two-sharer point was left in combA by THIS thread before the staging (plain load), that of a point with
// more sharers by the launch's first workgroups (coherent loads), and the freeze flag goes straight to its send slots (exchange
// F, SM.C:2374); !spTile, the regular tiles: shared points are skipped -- their own tiles do them
template <bool FINAL, int T, int MODE = 0>
__device__ __forceinline__ void smoothPoint(const MeshView& m, const State& s, const Prm& prm, const SmoothTileView& g, const SmoothLds& L,
                                            const SmoothRow& R, const SmoothTileMeta& tm, int tid, double& dist, int& fcount, bool spTile = false) {
    const double *cx = L.cx, *cy = L.cy, *cz = L.cz, *nx = L.nx, *ny = L.ny, *nz = L.nz;
    const bool mine = R.mine;
    const int p = R.p, selfL = R.selfL, wn4 = R.wn4, wc4 = R.wc4;
    const ushort4 *ppRow = R.ppRow, *pcRow = R.pcRow;
    const ushort4 pp0 = R.pp0, pp1 = R.pp1, pc0 = R.pc0, pc1 = R.pc1;
    if ((SMGPU_ABLATE & 16) && !spTile) { if (mine) stv(FINAL ? s.ptsNext : s.prop, p, ldsv(nx, ny, nz, selfL)); return; }
    if (mine && !(MODE == 2 && !spTile && R.slot >= 0)) {
        const unsigned fl = R.fl;
        const bool internal = fl & PF_INTERNAL;
        const V3 cur = ldsv(nx, ny, nz, selfL);
        V3 sum = v3(0, 0, 0), r1, r2, r3;
        double m1 = 0.0, m2 = 0.0, m3 = 0.0;   // mag(closestPoint1..3), SM.C:509-510
        int count = 0, hc = 0;
        double shortestCur = SMGPU_GREAT;   // SM.C:621; min over ALL neighbours of the current edge lengths
        const int slot = (MODE == 1 || (MODE == 2 && !spTile)) ? -1 : R.slot;
        if (slot >= 0) {
            // (the combined record of the shared point, left by k_halo_combineA*.  Combining two-sharer points HERE from the own and
            // the received record was a knob until round 4 -- SMGPU_HALO_INLINE, measured slower in round 2 -- and is gone: its inlined
            // code cost the kernel registers whether the knob was on or not: 64 VGPRs (7 waves per SIMD) without it, 89 (5) in round 3,
            // 97 + spills (4) with the master fold's two selects per sync -- 34.1 / 36.7 / 43.9 us per launch on 100^3)
            {
                const double* r = s.combA + (size_t)slot * SMGPU_HALO_A_DOUBLES;
                double w[SMGPU_HALO_A_DOUBLES];
#pragma unroll
                for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) w[q] = (MODE == 2 && R.peer < 0) ? ldCoh(r + q) : r[q];
                sum = v3(w[0], w[1], w[2]);
                r1 = v3(w[3], w[4], w[5]); r2 = v3(w[6], w[7], w[8]); r3 = v3(w[9], w[10], w[11]);
                const long long pk = __double_as_longlong(w[12]);
                count = (int)(pk & 0xffffffffll);
                hc = (int)(pk >> 32);
            }
            m1 = mag(r1); m2 = mag(r2); m3 = mag(r3);
            SMGPU_ELL_FOREACH_PRE(pp0, pp1, ppRow, wn4, T, {
                (void)j;
                const double len = mag(cur - ldsv(nx, ny, nz, e & 0x7fff));
                if (len < shortestCur) shortestCur = len;
            })
        } else {
            if (internal || prm.bndOn) {   // SM.C:116-130 (boundary points too with doBoundarySmoothing)
                if (!(SMGPU_ABLATE & 8))
                SMGPU_ELL_FOREACH_PRE(pc0, pc1, pcRow, wc4, T, {
                    sum = sum + ldsv(cx, cy, cz, (SMGPU_ABLATE & 2) ? 0 : e);
                    count = j + 1;
                })
            }
            // SM.C:325-387 (stable top three; boundary points look at boundary neighbours only)
            Top3 t3;
            SMGPU_ELL_FOREACH_PRE(pp0, pp1, ppRow, wn4, T, {
                // getPointDistance(neigh, cCoords) = |cCoords - neigh|; the same value serves SM.C:626
                const V3 dv_ = cur - ldsv(nx, ny, nz, (SMGPU_ABLATE & 2) ? 0 : (e & 0x7fff));
                const double len = (SMGPU_ABLATE & 1) ? magSqr(dv_) : mag(dv_);
                if (len < shortestCur) shortestCur = len;
                t3.offer(len, j, e, internal || !(e & 0x8000));
            })
            if (!t3.has2()) { s.acc->err = 1; r1 = r2 = r3 = v3(0, 0, 0); }
            else {
                r1 = ldsv(nx, ny, nz, t3.w1 & 0x7fff) - cur;
                r2 = ldsv(nx, ny, nz, t3.w2 & 0x7fff) - cur;
                r3 = !t3.has3() ? v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT) : ldsv(nx, ny, nz, t3.w3 & 0x7fff) - cur;
                m1 = t3.l1; m2 = t3.l2; m3 = !t3.has3() ? mag(r3) : t3.l3;
                if (g.usePairShare) {
                    hc = (pairBits<T>(g, tm, R, tid, t3.k1()) >> t3.k2()) & 1;
                } else {
                    const int nb = m.ppOff[p];
                    hc = shareCell(m, m.ppPt[nb + t3.k1()], m.ppPt[nb + t3.k2()]) ? 1 : 0;
                }
            }
        }
        V3 np = cur;
        if (count) np = divByCount(sum, count);                        // SM.C:155-163
        const double blendFrac = arRatioLen(r1, r2, m1, m2, m3, hc != 0, internal);
        if (blendFrac > 0.0) {                                         // SM.C:580-590
            const V3 aCoords = cur + 0.5 * (r1 + r2);   // (r1 + r2) / 2.0: halving is exact either way
            np = (1.0 - blendFrac) * np + blendFrac * aCoords;
        }
        {                                                              // SM.C:722-745
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        if (prm.layersOn) np = layerTreat(s, prm, p, internal, cur, np);   // SM.C:2283-2305
        const bool deferBnd = prm.bndOn && !internal;   // k_bnd_fix finishes the boundary points (kernels_boundary.hpp)
        if (prm.bndOn && internal) {                                   // SM.C:2356
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        bool frozen = false;                                           // SM.C:611-648
        {
            // min over the neighbours of |np - x_q|: the correctly rounded square root is monotone, so the minimum of the roots
            // IS the root of the minimum of the squares -- one sqrt instead of one per neighbour, the same bits
            // (NaN squares are skipped as NaN lengths were; lengths not below the loop's start value GREAT never won)
            double minSqr = __builtin_inf();
            if (!(SMGPU_ABLATE & 4))
            SMGPU_ELL_FOREACH_PRE(pp0, pp1, ppRow, wn4, T, {
                (void)j;
                const double t2 = magSqr(np - ldsv(nx, ny, nz, (SMGPU_ABLATE & 2) ? 0 : (e & 0x7fff)));
                if (t2 < minSqr) minSqr = t2;
            })
            const double rootMin = sqrtExact(minSqr);
            const double shortestNew = (rootMin < SMGPU_GREAT) ? rootMin : SMGPU_GREAT;
            const double shortest = (shortestNew < shortestCur) ? shortestNew : shortestCur;
            if (prm.totalMinFreeze && (shortest < prm.minEdge)) frozen = true;
            else if ((shortestNew < prm.minEdge) && (shortestNew < shortestCur)) frozen = true;
        }
        if (deferBnd) stv(s.prop, p, np);
        else if (FINAL && slot >= 0) {
            // shared point (multi-rank): its freeze flag still has to be OR-ed over the ranks (SM.C:2374);
            // k_shared_fix finishes it after exchange F
            if (MODE == 2 && s.ownF) {      // the fix role of the SAME launch reads both (k_smooth_halo): coherent stores
                stvCoh(s.prop, p, np);
                __hip_atomic_store(s.ownF + slot, frozen ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else stv(s.prop, p, np);
            s.frozen[p] = frozen ? 1 : 0;
            if (MODE == 2) {                                                     // exchange F, SM.C:2374
                const int v = frozen ? 1 : 0;
                if (R.ndst == 1) { if (s.push.slotF) stPeer(s.push.slotF[R.dst0], v); else s.sendF[R.dst0] = v; }
                else
                    for (int k = s.sendOff[slot]; k < s.sendOff[slot + 1]; ++k) {
                        const int sl = s.sendSlots[k];
                        if (s.push.slotF) stPeer(s.push.slotF[sl], v); else s.sendF[sl] = v;
                    }
            } else if (s.inlinePackF)
                for (int k = s.sendOff[slot]; k < s.sendOff[slot + 1]; ++k) s.sendF[s.sendSlots[k]] = frozen ? 1 : 0;   // exchange F, SM.C:2374
        } else if (FINAL) {
            if (frozen || (!internal && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
            dist = mag(np - cur) / prm.maxStep;
            stv(s.ptsNext, p, np);
        } else {
            stv(s.prop, p, np);
            s.frozen[p] = frozen ? 1 : 0;
            if (s.stepSqr) s.stepSqr[p] = magSqr(np - cur);   // (k_apply_swap: the residual's term, SM.C:1546-1565)
        }
    }
}

template <bool FINAL, int T>
__global__ void __launch_bounds__(T) k_smooth_tile(MeshView m, State s, Prm prm, SmoothTileView g, const int* tileList,
                                                    int nLaunch, int xcdMap) {
    const int stopped = s.acc->stop;   // tested after the staging, see k_geom_tile
    const int li = launchTile(nLaunch, xcdMap);
    if (li < 0) return;
    staggerFirstRound((int)blockIdx.x, 256 * 6);
    extern __shared__ double lds[];
    const int tile = tileList ? ((const_int_ptr)tileList)[li] : li, tid = threadIdx.x;
    const SmoothTileMeta tm = loadTileMeta(g, tile);
    const SmoothLds L = smoothLds(lds, tm);
    const SmoothRow R = smoothStage<T>(m, s, g, tm, L, tid);
    if (stopped) return;
    __syncthreads();
    double dist = 0.0;
    int fcount = 0;
    smoothPoint<FINAL, T>(m, s, prm, g, L, R, tm, tid, dist, fcount);
    if (FINAL) blockPublish<T>(s, dist, fcount, tile);
}

// Exchange A of the multi-rank path (local partial sums and closest points of the shared points, SM.C:108-131, 325-387) on the
// smoothing tiles that hold shared points: the same staged gather as k_smooth_tile instead of k_halo_packA's per-point
// gathers from global memory (a chain of dependent loads per point).  Writes the own record and its copies in the send slots.
struct PackView {
    double* ownA; double* sendA;
    const int* sendOff; const int* sendSlots;
    int centroidAll;            // boundary point smoothing: boundary points gather cell centres too (SM.C:116)
};
// (the first nLBlocks workgroups, a multiple of 8, pack exchange L's records instead -- k_halo_packL's work without its launch)
// bid: the workgroup's index among the pack workgroups; COHC: see smoothStage (pack role of k_geom_halo); SP: g = the shared
// points' own tiles (every point of a tile is a shared point; slots and first send slot by tile position)
template <int T, bool COHC = false, bool SP = false>
__device__ __forceinline__ void packTileBody(const MeshView& m, const State& s, const SmoothTileView& g, const PackView& pk, const int* tileList, int nLaunch,
                                             int xcdMap, int bid) {
    const int li = launchTile(nLaunch, xcdMap, bid);
    if (li < 0) return;
    extern __shared__ double lds[];
    const int tile = tileList ? ((const_int_ptr)tileList)[li] : li, tid = threadIdx.x;
    const SmoothTileMeta tm = loadTileMeta(g, tile);
    const SmoothLds L = smoothLds(lds, tm);
    const SmoothRow R = smoothStage<T, COHC, SP ? 2 : 1>(m, s, g, tm, L, tid, SlotTabs{s.spSlot, s.spPeer, s.spDst0, s.spNDst});
    __syncthreads();
    if (!R.mine) return;
    const int p = R.p;
    const int slot = R.slot;
    if (slot < 0) return;
    const double *cx = L.cx, *cy = L.cy, *cz = L.cz, *nx = L.nx, *ny = L.ny, *nz = L.nz;
    const int wn4 = R.wn4, wc4 = R.wc4;
    const ushort4 *ppRow = R.ppRow, *pcRow = R.pcRow;
    const ushort4 pp0 = R.pp0, pp1 = R.pp1, pc0 = R.pc0, pc1 = R.pc1;
    const bool internal = R.fl & PF_INTERNAL;
    const V3 cur = ldsv(nx, ny, nz, R.selfL);
    V3 sum = v3(0, 0, 0), r1, r2, r3;
    int count = 0, hc = 0;
    if (internal || pk.centroidAll) {
        SMGPU_ELL_FOREACH_PRE(pc0, pc1, pcRow, wc4, T, {
            sum = sum + ldsv(cx, cy, cz, e);
            count = j + 1;
        })
    }
    Top3 t3;
    SMGPU_ELL_FOREACH_PRE(pp0, pp1, ppRow, wn4, T, {
        const double len = mag(cur - ldsv(nx, ny, nz, e & 0x7fff));
        t3.offer(len, j, e, internal || !(e & 0x8000));
    })
    if (!t3.has2()) { s.acc->err = 1; r1 = r2 = r3 = v3(0, 0, 0); }
    else {
        r1 = ldsv(nx, ny, nz, t3.w1 & 0x7fff) - cur;
        r2 = ldsv(nx, ny, nz, t3.w2 & 0x7fff) - cur;
        r3 = !t3.has3() ? v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT) : ldsv(nx, ny, nz, t3.w3 & 0x7fff) - cur;
        if (g.usePairShare) {
            hc = (pairBits<T>(g, tm, R, tid, t3.k1()) >> t3.k2()) & 1;
        } else {
            const int nb = m.ppOff[p];
            hc = shareCell(m, m.ppPt[nb + t3.k1()], m.ppPt[nb + t3.k2()]) ? 1 : 0;
        }
    }
    double rec[SMGPU_HALO_A_DOUBLES];
    rec[0] = sum.x; rec[1] = sum.y; rec[2] = sum.z;
    rec[3] = r1.x; rec[4] = r1.y; rec[5] = r1.z;
    rec[6] = r2.x; rec[7] = r2.y; rec[8] = r2.z;
    rec[9] = r3.x; rec[10] = r3.y; rec[11] = r3.z;
    rec[12] = __longlong_as_double(((long long)hc << 32) | (long long)(unsigned int)count);
    double* o = pk.ownA + (size_t)slot * SMGPU_HALO_A_DOUBLES;
#pragma unroll
    for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) o[q] = rec[q];
    const int kBeg = (SP && R.ndst == 1) ? 0 : pk.sendOff[slot], kEnd = (SP && R.ndst == 1) ? 1 : pk.sendOff[slot + 1];
    for (int k = kBeg; k < kEnd; ++k) {
        const int sl = (SP && R.ndst == 1) ? R.dst0 : pk.sendSlots[k];
        // the record goes where it is consumed: the peer's receive slot with the peer-store transport, else the send buffer
        if (s.push.slotA) {
            // 13 doubles = 104 bytes: every other slot starts 8 bytes off a 16-byte boundary -- six 16-byte stores and one
            // 8-byte store either way
            double* d = s.push.slotA[sl];
            if (((size_t)d & 8u) == 0) {
#pragma unroll
                for (int q = 0; q < 12; q += 2) stPeer2(d + q, rec[q], rec[q + 1]);
                stPeer(d + 12, rec[12]);
            } else {
                stPeer(d, rec[0]);
#pragma unroll
                for (int q = 1; q < 13; q += 2) stPeer2(d + q, rec[q], rec[q + 1]);
            }
            continue;
        }
        double* d = pk.sendA + (size_t)sl * SMGPU_HALO_A_DOUBLES;
#pragma unroll
        for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) d[q] = rec[q];
    }
}
template <int T>
__global__ void __launch_bounds__(T) k_pack_tile(MeshView m, State s, SmoothTileView g, PackView pk, const int* tileList, int nLaunch,
                                                  int xcdMap, PackLArgs la, int nLBlocks, unsigned tag) {
    if (s.acc->stop) return;
    if ((int)blockIdx.x < nLBlocks) haloPackLOf(s, la, (int)blockIdx.x * T + (int)threadIdx.x);
    else packTileBody<T>(m, s, g, pk, tileList, nLaunch, xcdMap, (int)blockIdx.x - nLBlocks);
    pushSignal(s.push, 0, tag);      // exchange A and L leave together (no-op without the peer-store transport)
}

// ---- multi-rank, constraints off: launches whose workgroups play several roles ---------------------------------------------------
// One rank of a decomposed run did, per iteration, geometry | pack | [exchange A] | combine | smoothing | packF | [exchange F] |
// shared-fix: four of the six kernels are small and latency bound (one rank of eight on 100^3: 143-153 us against 101 us for the
// same sub-domain alone).  Here the iteration is geometry+pack | [A] | combine+smoothing+packF | [F] | shared-fix, and all the
// halo work runs on tiles of the SHARED POINTS ONLY (State::sp*, smgpu_halo_configure: ~110 shared points per tile instead of the
// ~40 a regular smoothing tile near a processor patch holds -- a third of the workgroups and of the staging):
//  * k_geom_halo: [the geometry tiles with a shared point: coherent stores of their cell centres, one ticket per workgroup]
//    [a first batch of the other geometry tiles] [the pack role on the shared points' tiles: waits for the ticket count, stages
//    the cell centres with coherent loads] [the rest of the geometry tiles];
//  * k_smooth_halo: [combine of the points with more than two sharers, 16 lanes per point] [the shared points' tiles: wait for the
//    peers' records where the transport needs it, two-sharer points combined by their own thread, freeze flags straight into the
//    send slots / the peers' receive slots] [ALL regular tiles, skipping their shared points: no dependence on the exchange].
// A halo workgroup is a chain of ~10 dependent round trips: at the tail of a launch it costs its full length (first version,
// measured: 74 + 50 us against 51 + 40 us for the plain kernels), in front of the bulk it runs next to it -- but every such
// workgroup holds a slot sized for the launch's fattest role, so there must be few of them (770 pack workgroups on the regular
// tiles in second place: 111 us).
// Workgroups are dispatched in index order per XCD, so a role only ever waits for workgroups that were dispatched before it;
// every wait is bounded (roleWait / pushWait raise Accum::err).
struct HaloG {
    const int* geomS; int nGeomS; const int* geomI; int nGeomI;   // geometry tiles with / without a shared point
    int nI1;                                                      // ... of the latter, how many go in front of the pack role
    int nPack;                                                    // the shared points' tiles
    unsigned* ticket; unsigned serial;                            // "the first role's workgroups of THIS launch are done" (roleDone / roleWait)
    unsigned tagA;
    int debug;                                                    // measurement aid (SMGPU_HALO_DEBUG): 1 = pack role returns at once, 2 = does not wait
};
template <int T, bool ORG>
__global__ void __launch_bounds__(T, (SMGPU_GEOM_WAVES * T) / 256) k_geom_halo(MeshView m, State s, GeomTileView g, int writeFaces, HaloG hg, int xcdMap, int deferN,
                                                                              int deferIter, double* deferLocal, double* deferHist, SmoothTileView sg, PackView pk) {
    const int bid = (int)blockIdx.x, gS = tileGrid(hg.nGeomS, xcdMap), g1 = tileGrid(hg.nI1, xcdMap), gP = tileGrid(hg.nPack, xcdMap);
    if (bid >= gS + g1 && bid < gS + g1 + gP) {
        if (s.acc->stop || (hg.debug & 1)) return;
        __builtin_amdgcn_s_setprio(3);      // (a chain of dependent steps the exchange waits for: issue priority over the tiles' waves)
        if (!(hg.debug & 2)) roleWait(hg.ticket, hg.serial, &s.acc->err);
        packTileBody<T, true, true>(m, s, sg, pk, nullptr, hg.nPack, xcdMap, bid - gS - g1);
        pushSignal(s.push, 0, hg.tagA, (unsigned)gP);      // exchange A leaves (peer-store transport)
        return;
    }
    // the three geometry roles through ONE copy of the tile code (see geomCell)
    const bool first = bid < gS, second = !first && bid < gS + g1;
    const int* list = first ? hg.geomS : (second ? hg.geomI : hg.geomI + hg.nI1);
    const int n = first ? hg.nGeomS : (second ? hg.nI1 : hg.nGeomI - hg.nI1);
    const int b = first ? bid : (second ? bid - gS : bid - gS - g1 - gP);
    geomTileBody<T, ORG>(m, s, g, 0, writeFaces, list, n, xcdMap, first ? deferN : 0, deferIter, deferLocal, deferHist, b, first);
    if (first) roleDone(hg.ticket, hg.serial, (unsigned)gS >> 3);      // (every workgroup of the role, padding included)
}

struct HaloFix {                             // the fix role of k_smooth_halo (k_shared_fix's work inside the launch), or nFix = 0
    int nFix;                                // workgroups (a multiple of 8; nShared / T of them with work)
    int nShared; const int* sharedLocal; const int* combOff; const int* combSlots; const int* recvF; int partialBase;
    PushWait pwF;
    unsigned* ticket; unsigned serial;       // "the shared points' role of THIS launch is done" (roleDone / roleWait)
    int sysLoads;
};
struct HaloS {
    int nTiles;                              // regular smoothing tiles (all of them; they skip their shared points)
    int nA;                                  // ... of which this many go in front of the fix role
    int nSp;                                 // the shared points' tiles
    int gM, nMultiBlocks, nMulti;            // first role: gM workgroups (a multiple of 8), nMultiBlocks of them with work
    const int* multiIdx; const int* multiSlots; double* combA;
    unsigned* ticket; unsigned serial;       // "the first role's workgroups of THIS launch are done" (roleDoneSmall / roleWait)
    PushWait pwA; unsigned tagF;
};
// SM.C:2374-2392 for the shared points inside the smoothing launch (k_shared_fix's work, see there): after the shared points'
// role of this launch (their proposals and local freeze flags: coherent loads) and exchange F (the peers' flags / the exchange
// stream's word), OR the flags, restore / count, write the new coordinates, publish the residual partials
template <int T>
__device__ __forceinline__ void haloFixBody(const MeshView& m, const State& s, const Prm& prm, const HaloFix& hf, int blk) {
    roleWait(hf.ticket, hf.serial, &s.acc->err);
    pushWait(hf.pwF);
    const int i = blk * T + (int)threadIdx.x;
    double dist = 0.0;
    int fcount = 0;
    if (i < hf.nShared) {
        const int p = hf.sharedLocal[i];
        int frz = __hip_atomic_load(s.ownF + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = hf.combOff[i]; k < hf.combOff[i + 1]; ++k) {
            const int sl = hf.combSlots[k];
            if (sl >= 0) frz |= hf.sysLoads ? ldSys(hf.recvF + sl) : hf.recvF[sl];
        }
        s.frozen[p] = frz ? 1 : 0;
        const uint8_t fl = m.pflags[p];
        const V3 cur = ldv(s.ptsCur, p);
        V3 np = ldvCoh(s.prop, p);
        if (frz || (!(fl & PF_INTERNAL) && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
        dist = mag(np - cur) / prm.maxStep;
        stv(s.ptsNext, p, np);
    }
    blockPublish<T>(s, dist, fcount, hf.partialBase + blk);
}
template <int T>
__global__ void __launch_bounds__(T) k_smooth_halo(MeshView m, State s, Prm prm, SmoothTileView g, SmoothTileView sg, HaloS hs, int xcdMap, HaloFix hf) {
    const int bid = (int)blockIdx.x, tid = threadIdx.x;
    if (bid < hs.gM) {
        if (bid >= hs.nMultiBlocks) return;
        __builtin_amdgcn_s_setprio(3);
        pushWait(hs.pwA);
        combineMulti<true>(bid, hs.nMulti, hs.multiIdx, hs.multiSlots, s.ownA, s.recvA, hs.combA, s.ownFold);
        roleDoneSmall(hs.ticket, hs.serial, (unsigned)hs.nMultiBlocks);
        return;
    }
    extern __shared__ double lds[];
    const int gSp = tileGrid(hs.nSp, xcdMap), gA = tileGrid(hs.nA, xcdMap);
    const bool spTile = bid < hs.gM + gSp;                  // (workgroup-uniform)
    if (!spTile && bid >= hs.gM + gSp + gA && bid < hs.gM + gSp + gA + hf.nFix) {
        const int blk = bid - hs.gM - gSp - gA;
        __builtin_amdgcn_s_setprio(3);
        if (blk * T < hf.nShared) haloFixBody<T>(m, s, prm, hf, blk);
        return;
    }
    if (spTile) __builtin_amdgcn_s_setprio(3);
    const SmoothTileView v = pickView(spTile, sg, g);
    // the regular tiles in two batches, in front of and behind the fix role
    const bool batchA = bid < hs.gM + gSp + gA;
    int li = spTile ? launchTile(hs.nSp, xcdMap, bid - hs.gM)
                    : (batchA ? launchTile(hs.nA, xcdMap, bid - hs.gM - gSp) : launchTile(hs.nTiles - hs.nA, xcdMap, bid - hs.gM - gSp - gA - hf.nFix));
    if (li < 0 && !spTile) return;
    if (!spTile && !batchA) li += hs.nA;
    const int tile = li >= 0 ? li : 0;      // (a padding workgroup of the shared points' role stages the first tile in vain: it still signals)
    const SmoothTileMeta tm = loadTileMeta(v, tile);
    const SmoothLds L = smoothLds(lds, tm);
    if (spTile) {
        // exchange A's combine for the thread's own point, FIRST -- while nothing else is live in registers (the two ranks' records
        // are 52 VGPRs: inside smoothPoint they took the kernel from 64 to 97) -- with the master's fold of the three sequential
        // syncs (combineTwoToMemory, SM.C:391-478); the record goes to combA and comes back in smoothPoint
        const int pi = tm.ptBeg + ((tid < tm.nPts) ? tid : 0);
        const int slot = (li >= 0 && tid < tm.nPts) ? s.spSlot[pi] : -1, peer = (slot >= 0) ? s.spPeer[pi] : -1;
        pushWait(hs.pwA);
        if (peer >= 0)
            combineTwoToMemory(s.ownA + (size_t)slot * SMGPU_HALO_A_DOUBLES, s.recvA + (size_t)(peer & 0x3fffffff) * SMGPU_HALO_A_DOUBLES,
                               (peer & 0x40000000) != 0, s.ownFold, hs.combA + (size_t)slot * SMGPU_HALO_A_DOUBLES);
    }
    // both kinds of tiles through ONE copy of the staging and of the per-point code (see geomCell)
    const SlotTabs tb{spTile ? s.spSlot : s.posSlot, spTile ? s.spPeer : nullptr, s.spDst0, s.spNDst};
    const SmoothRow R = smoothStage<T, false, 2>(m, s, v, tm, L, tid, tb);
    if (spTile && hs.nMultiBlocks > 0) roleWait(hs.ticket, hs.serial, &s.acc->err);
    __syncthreads();
    double dist = 0.0;
    int fcount = 0;
    if (li >= 0) smoothPoint<true, T, 2>(m, s, prm, v, L, R, tm, tid, dist, fcount, spTile);
    if (spTile) {
        if (hf.nFix > 0) roleDone(hf.ticket, hf.serial, (unsigned)gSp >> 3);      // the fix role of this launch may read the proposals
        pushSignal(s.push, 1, hs.tagF, (unsigned)gSp);      // exchange F leaves (peer-store transport / flagged arrangement)
    } else blockPublish<T>(s, dist, fcount, tile);
}

}  // namespace smgpu
