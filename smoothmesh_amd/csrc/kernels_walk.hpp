// kernels_walk.hpp -- compacted form of the face-angle freeze walk (SM.C:1347-1434) for meshes where
// many points lie outside the good angle range (e.g. refinement interfaces: coplanar face pairs give
// face angles of 180 degrees in every iteration).
//
// The walk itself is inherently sequential (LIFO stack, reads and writes isFrozenPoint as it goes), but
// everything expensive in it is a pure function of (current coordinates, proposals).  So:
//   1. k_walk_count / k_walk_fill : ordered compaction of the active points (ascending
//      point id) and of their pointPoints rows into dense tables;
//   2. k_walk_pred : one thread per table entry evaluates the geometric predicates in parallel
//      (self-deterioration, and for every neighbour "its move hurts me" with me at my proposal / at my
//      current position) -- the same bits as k_fa_pred;
//   3. the stack order is replayed over the bit tables.  Only points that can act (a true self bit or a true
//      neighbour bit) take part.  Early in a run the interaction graph falls into thousands of tiny components, but on
//      a refinement interface they percolate within ~10 iterations (10 M-cell cavity mesh, iteration 40: one component
//      with 116 k of the 136 k acting points), so "one lane per component" is no way out.  What makes the walk
//      parallel is its causal structure.  Number the first visits t = 0, 1, ... (descending point id) and let T(p) be
//      the step at which p gets frozen.  A visit at step t only acts on what earlier steps left behind, re-visits happen
//      within the step that caused them (LIFO), and the flags only ever go from 0 to 1, so (SM.C:1376-1433)
//        T(p) = min( -1 if frozen before the walk,
//                    t_p if p moves and its own move deteriorates its angles,                        (self freeze, :1391)
//                    T(q)  over entries q -> p that hold with q at its CURRENT position,               (re-visit of q, :1431)
//                    t_q   over such entries of a q that never moves or was frozen before the walk,    (first visit of q)
//                    t_q   over entries q -> p that hold with q at its PROPOSAL, if T(q) >= t_q )       (q still free at its visit)
//      Only the last rule is not monotone, and it looks strictly into the past.  So: fix the set A of points whose
//      proposal-state entries fire, solve the (monotone) rest by min-propagation, re-derive A from the T found, and
//      repeat until A stands.  Each round gets the earliest wrong decision -- and everything before it -- right, so
//      it terminates with exactly the sequential result (3 rounds of 5-19 propagation sweeps on the cavity meshes).
//      k_walk_fix runs all of it in ONE persistent launch with its own grid barrier; no copy, no host synchronisation.
//      The host replay of round 1 (one core, three stream synchronisations and two copies per iteration) stays
//      selectable as the A/B reference (SMGPU_WALK=host);
//   4. k_walk_apply marks the points the host replay froze (k_walk_fix writes the flags itself).
#pragma once
#include "kernels.hpp"

namespace smgpu {

// One 16-byte record per header / entry, so that a re-visit (a jump to a random point) touches one or two
// cache lines on the host.  The bits are laid out so that the host treats headers and entries alike:
//   act = !frozen[slot] && (bits & sel)          sel = 1: the visiting point sits at its proposal, 2: at its current position
//   bit0 / bit1  entry: the neighbour must freeze when the visiting point is at its proposal / current position
//                header: both = the point freezes itself when visited unfrozen (moved && own angles deteriorate)
//   bit4  entry: push the neighbour for a re-visit when it gets frozen (it has a slot and its re-visit can act)
//   bit5  `slot` is a real slot (its frozen flag is updated); otherwise slot = nRelevant + (frozen before the walk)
//   bit6  header: visited unfrozen and not freezing itself, the point acts from its proposal (sel = 1)
//   bit7  header
struct WalkItem {
    int slot;             // header: the point's relevant slot;   entry: the neighbour's relevant slot, or see bit5
    int id;               // header: the point id;                entry: the neighbour's point id
    unsigned bits;
    int hpos;             // position of the header item of `slot` (bit5 set)
};

struct WalkView {
    int* activeSlot;      // [P] slot of an active point, -1 otherwise
    int* blkA; int* blkE; // per 256-point block: active count / entry count, then exclusive offsets
    int* header;          // {nActive, nEntries}
    int* actIds;          // [nActive] point ids ascending
    int* actEntOff;       // [nActive+1]
    uint8_t* actBits;     // bit0 self move deteriorates, bit1 moved, bit2 frozen before the walk
    int* entOwner;        // [nEntries] slot of the point the entry belongs to
    int* entNbr;          // neighbour point id
    int* entSlot;         // neighbour's slot or -1
    uint8_t* entBits;     // bit0 N(moved), bit1 N(frozen), bit2 neighbour moving, bit3 neighbour frozen before the walk
    // second compaction: only points that can act (a true self bit or a true neighbour bit) and only their
    // true entries travel to the host; every other active point is a pure sink in the walk
    int* relSlot;         // [nActive] slot among the relevant points or -1
    int* header2;         // {nRelevant, nBadEntries}
    uint8_t* relBits;     // [nRelevant] actBits | bit3: a re-visit (point held at its current position) can act
    // The relevant points and their true entries as ONE item sequence in the order the reference's walk first
    // visits them (points by descending id, entries by ascending list position): per point a header item
    // followed by its entry items.  The host replays it with a flat, branch-free loop.
    int* hdrPos;          // [nRelevant] position of the point's header item
    WalkItem* items;      // [nRelevant + nBadEntries]
};

// device replay (k_walk_fix): state over the relevant slots
struct FixView {
    int* T;               // [nRelevant] freeze step (header position of the visit that froze the point), -1 before the walk, INT_MAX never
    int* act;             // [nRelevant] 1: the point is still free at its first visit and acts from its proposal
    unsigned* bar;        // grid barrier: two 64-bit counters used alternately (fixSync)
    int* flags;           // [3] the abort word of a barrier that timed out, [8..15] debug statistics
    uint8_t* actPrev;     // [nPoints] `act` of the point in the last walk it took part in: the first guess of the next one
};

// Ordered compaction of the active points and of their pointPoints rows: chunks of 4 096 points per workgroup, count launch +
// fill launch, no scan launch in between (the chunk* helpers of kernels.hpp).  k_walk_fill leaves {nActive, nEntries, 0} in
// w.header.  activeSlot[p] is only written for active points: it is valid iff faActive[p] carries this iteration's tag
// (activeSlotOf) -- no 4-byte store per mesh point and iteration.
// (StarCache, below: count[1] / count[2] say whether this iteration has a point without a record / a point the full pool could not take
// -- the two launches that serve those leave at once otherwise)
__global__ void __launch_bounds__(kBlock) k_walk_count(MeshView m, State s, WalkView w, int* starNeed = nullptr) {
    if (s.acc->stop) return;
    if (starNeed && blockIdx.x == 0 && threadIdx.x == 0) { starNeed[0] = 0; starNeed[1] = 0; }
    const int base = blockIdx.x * kChunk + threadIdx.x * kChunkPer;
    const uint4 q = chunkMarks(s.faActive, base, m.nPoints);
    int a = 0, e = 0;
    if (q.x | q.y | q.z | q.w) {
        for (int i = 0; i < kChunkPer; ++i)
            if (chunkByte(q, i) == s.faGen && base + i < m.nPoints) { ++a; e += m.ppOff[base + i + 1] - m.ppOff[base + i]; }
    }
    chunkReduce2(a, e);
    if (threadIdx.x == 0) { w.blkA[blockIdx.x] = a; w.blkE[blockIdx.x] = e; }
}
__device__ __forceinline__ int activeSlotOf(const State& s, const WalkView& w, int q) { return (s.faActive[q] == s.faGen) ? w.activeSlot[q] : -1; }

__global__ void __launch_bounds__(kBlock) k_walk_fill(MeshView m, State s, WalkView w, const int* starSlot = nullptr, int* starNeed = nullptr) {
    if (s.acc->stop) return;
    __shared__ uint16_t list[kChunk];
    const int cb = blockIdx.x * kChunk, base = cb + threadIdx.x * kChunkPer;
    const int n = chunkCompact(chunkMarks(s.faActive, base, m.nPoints), s.faGen, base, m.nPoints, list);
    int pa, pe, ta, te;
    chunkPrefix2(w.blkA, w.blkE, (int)blockIdx.x, (int)gridDim.x, false, pa, pe, ta, te);
    int running = 0;
    for (int r = 0; r < n; r += kBlock) {              // one active point per thread
        const int i = r + threadIdx.x;
        const int p = (i < n) ? cb + list[i] : -1;
        const int nb = (p >= 0) ? m.ppOff[p] : 0, deg = (p >= 0) ? m.ppOff[p + 1] - nb : 0;
        int xe, tot;
        chunkScan1(deg, xe, tot);
        if (p >= 0) {
            const int slot = pa + i, eo = pe + running + xe;
            w.activeSlot[p] = slot;
            w.actIds[slot] = p;
            w.actEntOff[slot] = eo;
            if (starSlot) { const int sl = starSlot[p]; if (sl == -1) starNeed[0] = 1; else if (sl == -3) starNeed[1] = 1; }
            for (int j = 0; j < deg; ++j) { w.entOwner[eo + j] = slot; w.entNbr[eo + j] = m.ppPt[nb + j]; }
        }
        running += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        w.header[0] = pa + n; w.header[1] = pe + running; w.header[2] = 0;   // [2]: slots left by k_walk_pred_star
    }
}

// calcMinMaxFaceAngleForPoint (SM.C:1276-1308; pointFaceAngles in kernels.hpp) spread over the 32 lanes of a half wave:
// the (edge of p, cell around that edge) pairs -- 24 for an interior hex point -- are dealt to the lanes, each lane forms
// its cell's angle exactly as edgeFaceAngles<true> does (two projected face-centre vectors with the substituted
// coordinates, the projected cell centre, two acos), and min / max are reduced over the lanes (order-free).  One thread
// walking all of them was a chain of ~200 dependent gathers; a lane's chain is ~10.  All 32 lanes must call it with the
// same arguments.
__device__ __forceinline__ void groupPointFaceAngles(const MeshView& m, const State& s, int p, const V3& c1, int i2, const V3& c2,
                                                     int lane, double& mn, double& mx) {
    const int eb = m.ppOff[p], nEdgesP = m.ppOff[p + 1] - eb;
    if (nEdgesP > 32) {   // extreme valence: one lane does it the plain way
        if (lane == 0) pointFaceAngles(m, s, p, c1, i2, c2, mn, mx);
        mn = __shfl(mn, 0, 32);
        mx = __shfl(mx, 0, 32);
        return;
    }
    const int myE = (lane < nEdgesP) ? m.peEdge[eb + lane] : -1;
    const int myN = (myE >= 0) ? m.ecOff[myE + 1] - m.ecOff[myE] : 0;
    int incl = myN;
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(incl, o, 32);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 31, 32);
    mn = 2.0 * SMGPU_PI;
    mx = 0.0;
    for (int base = 0; base < total; base += 32) {
        const int idx = base + lane;
        int e = -1, first = 0;
        for (int j = 0; j < nEdgesP; ++j) {      // the edge whose cells include pair idx
            const int hi = __shfl(incl, j, 32), ej = __shfl(myE, j, 32), nj = __shfl(myN, j, 32);
            if (e < 0 && idx < hi) { e = ej; first = hi - nj; }
        }
        if (idx >= total) continue;
        const int k = m.ecOff[e] + (idx - first);
        // edgeFaceAngles<true> for the one cell k of edge e
        const int e0I = m.edges[2 * e], e1I = m.edges[2 * e + 1];
        V3 e0 = ldv(s.ptsCur, e0I), e1 = ldv(s.ptsCur, e1I);
        if (e0I == p) e0 = c1; else if (i2 >= 0 && e0I == i2) e0 = c2;
        if (e1I == p) e1 = c1; else if (i2 >= 0 && e1I == i2) e1 = c2;
        const V3 cC = 0.5 * (e0 + e1);
        const V3 d = e1 - e0;
        const V3 eVec = d / mag(d);
        const int fb = m.efOff[e];
        auto faceVec = [&](int f) -> V3 {
            V3 fc = v3(0, 0, 0);   // calcFaceCenter SM.C:1103-1130
            const int b = m.faceOff[f], n = m.faceOff[f + 1] - b;
            for (int i = 0; i < n; ++i) {
                const int q = m.facePts[b + i];
                if (q == p) fc = fc + c1;
                else if (i2 >= 0 && q == i2) fc = fc + c2;
                else fc = fc + ldv(s.ptsCur, q);
            }
            fc = divByCount(fc, n);   // = fc / double(n), bit for bit
            const V3 cf = cC - fc;
            const double dp = dot(cf, eVec);
            const V3 pC = fc + dp * eVec;
            const V3 w = pC - cC;
            return w / mag(w);
        };
        const V3 p0 = faceVec(m.efFace[fb + m.ecF0[k]]);
        const V3 p1 = faceVec(m.efFace[fb + m.ecF1[k]]);
        const V3 cc = ldv(s.cellCtr, m.ecCell[k]);  // mesh.C()[cellI] of the CURRENT mesh, SM.C:1218
        const V3 cf = cC - cc;
        const double dp = dot(cf, eVec);
        const V3 pC = cc + dp * eVec;
        const V3 w = pC - cC;
        const V3 cV = w / mag(w);
        const double angle = clampAcos(dot(p0, cV)) + clampAcos(dot(cV, p1));   // calcEdgeCenterEdgeAngle SM.C:980-998
        if (angle < mn) mn = angle;
        if (angle > mx) mx = angle;
    }
    for (int o = 16; o > 0; o >>= 1) {
        const double a = __shfl_xor(mn, o, 32), b = __shfl_xor(mx, o, 32);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
}

// predicates: k_walk_pred_self, one half wave per active point (self test), then k_walk_pred, one half wave per (active point,
// neighbour) entry.  nA < 0: the counts are read from the device header (no host read-back) and the groups stride over them.
// What no replay can ever consult is not evaluated (the decisions are the reference's, bit for bit):
//   * a neighbour frozen before the walk is skipped by every visit (SM.C:1411), so its entry needs no angles;
//   * "the neighbour's move hurts me at my PROPOSAL" (bit0) is only read when the point acts from its proposal, i.e. when it is
//     visited unfrozen and does not freeze itself (SM.C:1376-1399): never for a point whose own move deteriorates its angles
//     (it is frozen by its first visit at the latest) nor for one frozen before the walk -- on a refinement interface that is
//     ~88 % of the acting points, and half of each of their entries' work.
// onlyLeft: only the slots k_walk_pred_star could not take (their actBits carry kStarLeft) are evaluated
constexpr uint8_t kStarLeft = 0x80;
__global__ void __launch_bounds__(kBlock) k_walk_pred_self(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, int onlyLeft) {
    if (s.acc->stop) return;
    if (nA < 0) { nA = w.header[0]; nE = w.header[1]; }
    if (onlyLeft && w.header[2] == 0) return;
    const int lane = threadIdx.x & 31;
    for (int t = (blockIdx.x * kBlock + threadIdx.x) >> 5; t < nA; t += (gridDim.x * kBlock) >> 5) {
        if (t == 0 && lane == 0) w.actEntOff[nA] = nE;
        if (onlyLeft && !(w.actBits[t] & kStarLeft)) continue;
        const int p = w.actIds[t];
        const V3 cur = ldv(s.ptsCur, p);
        const V3 np = ldv(s.prop, p);
        const bool moved = (np != cur);
        const bool frozenBefore = s.frozen[p] != 0;
        uint8_t sb = (moved ? 2 : 0) | (frozenBefore ? 4 : 0);
        if (moved && !frozenBefore) {   // SM.C:1385-1394 (a frozen point is held at its current position: no self test)
            double mn, mx;
            groupPointFaceAngles(m, s, p, np, -1, np, lane, mn, mx);
            const double curMin = s.ptMin[p], curMax = s.ptMax[p];
            if (((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax))) sb |= 1;
            if (lane == 0) noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
        }
        if (lane == 0) w.actBits[t] = sb | (onlyLeft ? kStarLeft : 0);
    }
}
__global__ void __launch_bounds__(kBlock) k_walk_pred(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, int onlyLeft) {
    if (s.acc->stop) return;
    if (nA < 0) { nA = w.header[0]; nE = w.header[1]; }
    if (onlyLeft && w.header[2] == 0) return;
    const int lane = threadIdx.x & 31;
    for (int e = (blockIdx.x * kBlock + threadIdx.x) >> 5; e < nE; e += (gridDim.x * kBlock) >> 5) {
        const int slot = w.entOwner[e];
        if (onlyLeft && !(w.actBits[slot] & kStarLeft)) continue;
        const int p = w.actIds[slot];
        const int q = w.entNbr[e];
        const V3 nq = ldv(s.prop, q);
        const bool qFrozen = s.frozen[q] != 0;
        uint8_t nb = qFrozen ? 8 : 0;
        if (!qFrozen && nq != ldv(s.ptsCur, q)) {   // SM.C:1411-1414: the neighbour is free and moving
            nb |= 4;
            const V3 cur = ldv(s.ptsCur, p);
            const double curMin = s.ptMin[p], curMax = s.ptMax[p];
            const uint8_t sb = w.actBits[slot] & 0x7f;
            double mn, mx;
            groupPointFaceAngles(m, s, p, cur, q, nq, lane, mn, mx);   // this point held at its current position
            const bool badF = ((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax));
            if (lane == 0) noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
            if (badF) nb |= 2;
            if (!(sb & 2)) { if (badF) nb |= 1; }           // not moved: proposal = current position
            else if (!(sb & 5)) {                           // acts from its proposal if it is still free at its first visit, SM.C:1419
                const V3 np = ldv(s.prop, p);
                groupPointFaceAngles(m, s, p, np, q, nq, lane, mn, mx);
                if (((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax))) nb |= 1;
                if (lane == 0) noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
            }
        }
        if (lane == 0) { w.entBits[e] = nb; w.entSlot[e] = activeSlotOf(s, w, q); }
    }
}

// ---- the predicates of one active point by one wave, from its "star" staged in LDS ----------------------------------------
// Every angle the walk can ask about point p involves only the faces that contain p (the rings of p's edges), their vertices
// and the centres of p's cells.  The per-entry kernels above fetch that neighbourhood again for every (entry, state, edge, cell)
// -- ~2 000 gathers of 24 bytes per active point.  Here a wave stages the star once (faces of p with their vertex ids and
// current coordinates: ~48 vertex slots for an interior hex point) and every lane keeps ITS place in the ring of one of p's
// edges for all jobs of the point: ring face i (as a star-local id) and the cell between ring faces i and i + 1 (its centre).
// A job -- self test, or one entry in one state -- is: every lane forms the projected centre vector of its face and of its
// cell with the substituted coordinates read from LDS (edgeFaceAngles<true>'s arithmetic, operation for operation), takes the
// next ring face's vector from its neighbour lane, adds the two acos (SM.C:980-998; the sum does not depend on which of the
// cell's two faces comes first), and min / max are reduced over the half wave.  Each face vector is thus formed once per
// job, not once per adjacent cell.  Two jobs run side by side on the two halves.  Points whose star exceeds the caps, or
// with a non-manifold edge ring, are left (kStarLeft) to the gather form above.
constexpr int kStarFaces = 32, kStarVerts = 128, kStarEnts = 32;
// Round 4: what a lane needs in EVERY job but does not change between jobs lives in LDS, not in registers -- the half wave's point
// (current / proposed coordinates, its two current angle bounds), the entries' neighbours with their proposals, the coordinates of
// the lane's own edge neighbour (a vertex slot of the star) -- and a vertex slot carries a one-byte ROLE (the point itself, entry
// e's neighbour, anybody else) instead of a point id to compare.  Before, those 28 registers per lane pushed the kernel over its
// budget of 128: 27 VGPRs spilled, 104 bytes of scratch per lane, written once per point and re-read inside the job loop (272 MB of
// scratch writes per launch on the 10 M-cell cavity mesh).
constexpr unsigned char kRoleSelf = 0xFF, kRoleOther = 0xFE, kRoleNoEntry = 0xFD;
template <int NV>
struct StarLdsT {
    int fid[kStarFaces];
    int voff[kStarFaces + 1];
    unsigned char role[NV];              // kRoleSelf: the point itself, e < kStarEnts: the neighbour of entry e, kRoleOther
    double vx[NV], vy[NV], vz[NV];
    unsigned char nb[kStarEnts];
    signed char job[2 * kStarEnts + 2];  // the jobs that are needed, in order: -1 = the self test, e = entry e with p at its current
                                         // position, 64 + e = entry e with p at its proposal
    double pc[3], pn[3], pMin, pMax;     // the half wave's point: current and proposed coordinates, ptMin / ptMax
    int eq[kStarEnts];                   // entry e's neighbour (point id)
    double ex[kStarEnts], ey[kStarEnts];
    union {
        double ez[kStarEnts];            // (ex, ey, ez): entry e's neighbour at its proposal
        struct { int fbeg[kStarFaces]; unsigned char vface[NV]; } st;   // staging only: first entry of the face in facePts,
                                                                                 // the star-local face of every vertex slot
    };
};
typedef StarLdsT<kStarVerts> StarLds;
static_assert(sizeof(StarLds) * 8 <= 40960, "k_walk_pred_star: four workgroups per CU need <= 40 KB of LDS each");
struct StarLane {          // a lane's place: ring position i of edge (p, xI)
    bool valid, hasCell;
    int xEnt;              // the edge's other end point as an entry of the point (pointPoints order = entry order)
    int xSlot;             // ... and as a vertex slot of the star (its current coordinates)
    bool pFirst;           // p is the edge's first end point (edges[2e])
    V3 cc;                 // centre of the cell between ring faces i and i + 1
    int l;                 // ring face i as a star-local id
    int nextLane;          // (within the half wave) the lane of ring face i + 1
};
// the pair's angle for the lanes with a cell (valid && hasCell), anything for the others; all 32 lanes of the half call it.
// c1 = where the point is in this job, ei = the entry whose neighbour sits at its proposal c2 (kRoleNoEntry: nobody else moves)
__device__ __forceinline__ double starLaneAngle(const StarLds& L, const StarLane& P, const V3& c1, int ei, const V3& c2) {
    V3 fv = v3(0, 0, 0), cV = v3(0, 0, 0);
    if (P.valid) {
        const V3 xs = sel3(P.xEnt == ei, c2, v3(L.vx[P.xSlot], L.vy[P.xSlot], L.vz[P.xSlot]));
        const V3 e0 = sel3(P.pFirst, c1, xs), e1 = sel3(P.pFirst, xs, c1);
        const V3 cC = 0.5 * (e0 + e1);
        const V3 d = e1 - e0;
        const V3 eVec = d / mag(d);
        {
            V3 fc = v3(0, 0, 0);   // calcFaceCenter SM.C:1103-1130
            const int b = L.voff[P.l], n = L.voff[P.l + 1] - b;
            for (int i = 0; i < n; ++i) {
                const int r = L.role[b + i];
                if (r == kRoleSelf) fc = fc + c1;
                else if (r == ei) fc = fc + c2;
                else fc = fc + v3(L.vx[b + i], L.vy[b + i], L.vz[b + i]);
            }
            fc = divByCount(fc, n);   // = fc / double(n), bit for bit
            const V3 cf = cC - fc;
            const double dp = dot(cf, eVec);
            const V3 pC = fc + dp * eVec;
            const V3 w = pC - cC;
            fv = w / mag(w);
        }
        if (P.hasCell) {
            const V3 cf = cC - P.cc;
            const double dp = dot(cf, eVec);
            const V3 pC = P.cc + dp * eVec;
            const V3 w = pC - cC;
            cV = w / mag(w);
        }
    }
    const V3 fn = v3(__shfl(fv.x, P.nextLane, 32), __shfl(fv.y, P.nextLane, 32), __shfl(fv.z, P.nextLane, 32));
    return clampAcos(dot(fv, cV)) + clampAcos(dot(cV, fn));   // calcEdgeCenterEdgeAngle SM.C:980-998
}
// min / max of the lanes' angles over a half wave; lanes without a cell contribute the neutral values
__device__ __forceinline__ void starReduce(bool valid, double angle, double& mn, double& mx) {
    mn = valid ? angle : 2.0 * SMGPU_PI;   // every angle is < 2 pi and > 0: the reference's start values never win
    mx = valid ? angle : 0.0;
    for (int o = 16; o > 0; o >>= 1) {
        const double a = __shfl_xor(mn, o, 32), b = __shfl_xor(mx, o, 32);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
}

#ifndef SMGPU_STAR_WAVES
#define SMGPU_STAR_WAVES 4   // waves per SIMD the register allocation aims at (measured on the 10 M-cell cavity mesh: 3 -> 945 us, 4 -> 833 us)
#endif
// TWO active points per wave, one per half wave (round 3).  A point's star needs <= 32 lanes for everything (edges, faces,
// ring places, entries), and what bounds the kernel besides FP64 issue is the chain of dependent gathers in front of the
// arithmetic (slot -> point -> rows -> faces -> vertices -> coordinates; edges -> ring -> cells -> centres: nine round trips
// at three waves per SIMD): with a point per half every load instruction of that chain serves two points.  The halves then
// run their own jobs, one per step (before: two jobs of ONE point side by side, the second half idle on the odd job).  All
// cross-lane traffic is 32 wide.
__global__ void __launch_bounds__(kBlock, SMGPU_STAR_WAVES) k_walk_pred_star(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, unsigned long long* opCount) {
    if (s.acc->stop) return;
    if (nA < 0) { nA = w.header[0]; nE = w.header[1]; }
    __shared__ StarLds lds[kBlock / 32];
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    StarLds& L = lds[threadIdx.x >> 5];
    const int groups = gridDim.x * (kBlock / 32);
    // the loop is wave-uniform (half 0 decides; half 1 of the last round may be without a point: live = false)
    for (int a0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2; a0 < nA; a0 += groups) {
        const int a = a0 + half;
        bool live = a < nA;
        if (a == 0 && hl == 0) w.actEntOff[nA] = nE;
        const int p = live ? w.actIds[a] : 0;
        bool moved;
        const bool frozenBefore = s.frozen[p] != 0;
        {   // the point itself goes to LDS (every job reads it from there)
            const V3 cur = ldv(s.ptsCur, p);
            const V3 np = ldv(s.prop, p);
            moved = (np != cur);
            if (hl == 0) {
                L.pc[0] = cur.x; L.pc[1] = cur.y; L.pc[2] = cur.z; L.pn[0] = np.x; L.pn[1] = np.y; L.pn[2] = np.z;
                L.pMin = s.ptMin[p]; L.pMax = s.ptMax[p];
            }
        }
        const int eBeg = live ? w.actEntOff[a] : 0, eEnd = live ? ((a + 1 < nA) ? w.actEntOff[a + 1] : nE) : 0, nEnt = eEnd - eBeg;
        // the point's edges with the lengths of their face rings, its faces with their vertex counts
        const int eb = m.ppOff[p], nEdgesP = live ? m.ppOff[p + 1] - eb : 0;
        const int fb = m.pfOff[p], nF = live ? m.pfOff[p + 1] - fb : 0;
        const int myE = (hl < nEdgesP) ? m.peEdge[eb + hl] : -1;
        const int myNf = (myE >= 0) ? m.efOff[myE + 1] - m.efOff[myE] : 0;
        const bool myRingBad = (myE >= 0) && !m.edgeRingOk[myE];
        const int myF = (hl < nF) ? m.pfFace[fb + hl] : -1;
        const int myFb = (myF >= 0) ? m.faceOff[myF] : 0;
        const int myV = (myF >= 0) ? m.faceOff[myF + 1] - myFb : 0;
        int inclN = myNf, inclV = myV;
        for (int o = 1; o < 32; o <<= 1) {
            const int tn = __shfl_up(inclN, o, 32), tv = __shfl_up(inclV, o, 32);
            if (hl >= o) { inclN += tn; inclV += tv; }
        }
        const int totalLanes = __shfl(inclN, 31, 32), totalV = __shfl(inclV, 31, 32);
        const unsigned ringBad = (unsigned)(__ballot(myRingBad) >> (32 * half));
        // (entry j = the neighbour across edge j: pointPoints and pointEdges share their offsets, so nEnt == nEdgesP)
        const bool fits = nEdgesP <= 32 && nF <= kStarFaces && totalLanes <= 32 && totalV <= kStarVerts && nEnt <= kStarEnts && nEnt == nEdgesP && ringBad == 0u;
        if (live && !fits && hl == 0) { w.actBits[a] = kStarLeft; atomicAdd(&w.header[2], 1); }   // left to k_walk_pred_self / k_walk_pred (onlyLeft)
        live = live && fits;
        // the entries of the point: lane i holds entry i (its neighbour, whether that one is free and moving)
        const int q = (live && hl < nEnt) ? w.entNbr[eBeg + hl] : -1;
        // Stage the star.  The loads are arranged in LEVELS of independent requests (what bounds this kernel besides FP64 issue
        // is the chain of dependent gathers, not their number): the edge lanes fetch their edge's cell range and end points
        // here, next to the faces' vertex ranges (the ring places below get them by shuffle instead of loading them one level
        // later), and the vertices are fetched by SLOT (four slots per lane: two round trips for the whole star) instead of by
        // a loop over each face's vertices (two dependent round trips per vertex).
        const int myCb = (myE >= 0) ? m.ecOff[myE] : 0, myNc = (myE >= 0) ? m.ecOff[myE + 1] - myCb : 0;
        const int myEfb = (myE >= 0) ? m.efOff[myE] : 0;
        const int myE0 = (myE >= 0) ? m.edges[2 * myE] : -1;
        if (live && hl < nF) {
            const int o = inclV - myV;
            L.fid[hl] = myF;
            L.voff[hl] = o;
            L.st.fbeg[hl] = myFb;
            for (int v = 0; v < myV; ++v) L.st.vface[o + v] = (unsigned char)hl;
        }
        if (live && hl == 0) L.voff[nF] = totalV;
        if (live && hl < nEnt) L.eq[hl] = q;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // this lane's ring place
        StarLane P;
        P.valid = live && hl < totalLanes;
        P.hasCell = false; P.xEnt = 0; P.xSlot = 0; P.pFirst = true; P.cc = v3(0, 0, 0); P.l = 0; P.nextLane = hl;
        int ringFaceId = -1, ringCellAt = -1;
        {
            int first = 0, nfj = 0, cb = 0, nc = 0, efb = 0, e0I = -1, jOf = 0;
            bool found = false;
            for (int j = 0; j < nEdgesP; ++j) {
                const int hi = __shfl(inclN, j, 32), nj = __shfl(myNf, j, 32);
                const int cbj = __shfl(myCb, j, 32), ncj = __shfl(myNc, j, 32), efbj = __shfl(myEfb, j, 32);
                const int e0j = __shfl(myE0, j, 32);
                if (!found && hl < hi) { found = true; first = hi - nj; nfj = nj; cb = cbj; nc = ncj; efb = efbj; e0I = e0j; jOf = j; }
            }
            if (P.valid) {
                const int i = hl - first;
                P.hasCell = i < nc;
                P.nextLane = (i + 1 < nfj) ? hl + 1 : first;          // closed ring: the last cell ends at face 0
                P.pFirst = (e0I == p);
                P.xEnt = jOf;
                ringFaceId = efb + i;
                if (P.hasCell) ringCellAt = cb + i;
            }
        }
        // level: vertex ids by slot, the ring's face and cell, the entry neighbour's two positions
        int vg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = hl + 32 * u;
            vg[u] = -1;
            if (live && k < totalV) { const int l = L.st.vface[k]; vg[u] = m.facePts[L.st.fbeg[l] + (k - L.voff[l])]; }
        }
        const int rf = (ringFaceId >= 0) ? m.ringFace[ringFaceId] : -1;
        const int rc = (ringCellAt >= 0) ? m.ringCell[ringCellAt] : -1;
        V3 nq = v3(0, 0, 0);
        bool eligible = false;
        unsigned char nb0 = 0;
        if (q >= 0) {
            nq = ldv(s.prop, q);
            const bool qFrozen = s.frozen[q] != 0;
            nb0 = qFrozen ? 8 : 0;
            eligible = !qFrozen && nq != ldv(s.ptsCur, q);   // SM.C:1411-1414
            if (eligible) nb0 |= 4;
        }
        // level: the coordinates; the role of every vertex slot (the point itself / entry e's neighbour / anybody else)
        V3 vc[4];
        unsigned char role[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            vc[u] = (vg[u] >= 0) ? ldv(s.ptsCur, vg[u]) : v3(0, 0, 0);
            role[u] = (vg[u] == p) ? kRoleSelf : kRoleOther;
        }
        if (rc >= 0) P.cc = ldv(s.cellCtr, rc);                      // mesh.C()[cellI] of the CURRENT mesh, SM.C:1218
        for (int e = 0; e < nEnt; ++e) {
            const int qe = L.eq[e];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (vg[u] == qe) role[u] = (unsigned char)e;
        }
        __builtin_amdgcn_wave_barrier();                              // (ez below overlays the staging tables read above)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = hl + 32 * u;
            if (vg[u] >= 0) { L.role[k] = role[u]; L.vx[k] = vc[u].x; L.vy[k] = vc[u].y; L.vz[k] = vc[u].z; }
        }
        if (q >= 0) { L.nb[hl] = nb0; L.ex[hl] = nq.x; L.ey[hl] = nq.y; L.ez[hl] = nq.z; }
        if (P.valid) for (int l = 0; l < nF; ++l) if (L.fid[l] == rf) P.l = l;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (P.valid) {       // the slot of the edge's other end point in this lane's ring face (it is a vertex of every face of the ring)
            for (int i = L.voff[P.l]; i < L.voff[P.l + 1]; ++i) if (L.role[i] == P.xEnt) P.xSlot = i;
        }
        const bool counts = P.valid && P.hasCell;
        // Jobs, one per step and half wave, ONE call site for all of them (a job's state is data, not control flow).  Job codes:
        // -1 the self test (p at its proposal), e = entry e with p at its current position, 64 + e = entry e with p at its
        // proposal.  Only jobs whose result can be consulted are listed: the self test of a point that moves and is still free,
        // the entries of the neighbours that are free and moving; the proposal-state jobs are appended after the first step,
        // once the self test has told whether p can act from its proposal at all (it is only read if p is still free at its
        // first visit and does not freeze itself, SM.C:1376-1399).
        // (Round 4, measured and removed: entries whose TOUCHED ring places are disjoint sharing a step -- the move of one
        // neighbour touches about half of a star's places, an untouched pair keeps its current angle resp. the self test's, so
        // opposite neighbours of a hex point can be evaluated side by side, exactly.  All tests bit-equal, but on the refinement
        // interfaces that make up the active set a point has 2-4 eligible neighbours and they are ADJACENT, their places overlap:
        // the colouring saved no steps -- 348 M against 328 M VALU wave instructions per launch on the 10 M-cell cavity mesh,
        // profiles/r4/pmc_k_walk_pred_star.txt -- and cost 5 %.)
        unsigned sbits = (moved ? 2u : 0u) | (frozenBefore ? 4u : 0u);
        const bool selfNeeded = live && moved && !frozenBefore;
        const unsigned elig = (unsigned)(__ballot(eligible) >> (32 * half));
        const int nEl = __popc(elig), first = selfNeeded ? 1 : 0;
        const int myRank = __popc(elig & ((1u << hl) - 1u));
        if (eligible) L.job[first + myRank] = (signed char)hl;
        if (hl == 0 && selfNeeded) L.job[0] = -1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#ifdef SMGPU_STAR_ABLATE_JOBS
        int nJobs = 0;      // (measurement build: staging only)
#else
        int nJobs = live ? first + nEl : 0;
#endif
        for (int j = 0; j < nJobs; ++j) {
            const int code = (int)L.job[j];
            const bool isSelf = code < 0, atProp = code >= 64;
            const int ei = isSelf ? 0 : (code & 63);                 // entry of this job
            const double* pc = (isSelf || atProp) ? L.pn : L.pc;     // where the point is in this job
            const V3 c1 = v3(pc[0], pc[1], pc[2]);
            const V3 c2 = v3(L.ex[ei], L.ey[ei], L.ez[ei]);          // (not used by the self test: no slot has the role kRoleNoEntry)
            const double angle = starLaneAngle(L, P, c1, isSelf ? (int)kRoleNoEntry : ei, c2);
            double mn, mx;
            starReduce(counts, angle, mn, mx);
            const double curMin = L.pMin, curMax = L.pMax;
            const bool isBad = ((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax));   // SM.C:1391-1399, 1421-1427
            if (hl == 0) noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
            if (j == 0) {
                if (selfNeeded && isBad) sbits |= 1u;                // (the reduction leaves mn / mx in every lane of the half)
                if (moved && !(sbits & 5u) && nEl > 0) {             // p can act from its proposal: the proposal-state jobs
                    if (eligible) L.job[nJobs + myRank] = (signed char)(64 + hl);
                    nJobs += nEl;
                }
            }
            if (!isSelf && hl == 0) {
                unsigned char v = L.nb[ei];
                if (atProp) { if (isBad) v |= 1; }
                else {
                    if (isBad) v |= 2;
                    if (isBad && !moved) v |= 1;                     // not moved: proposal = current position
                }
                L.nb[ei] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (live && hl == 0) w.actBits[a] = (uint8_t)sbits;
        if (opCount) {
            // timing passes only: the ALGORITHMIC FP64 instructions of this point's jobs, by the reference's arithmetic
            // (SM.C:1135-1231 per (edge, cell) pair: two projected face-centre vectors 77 + 3 (n - 1) + (3 | 33) each for a face
            // of n vertices, the projected cell centre 77, two clamped acos and their sum 95 -- sqrt = 22, division = 11 per
            // component, acos = 40; per edge of the point 69 for the edge vector) times the jobs that were needed.  One add per
            // point, spread over 64 words.
            const int nv = counts ? L.voff[P.l + 1] - L.voff[P.l] : 0;
            int ops = counts ? 2 * (77 + 3 * (nv - 1) + (((nv & (nv - 1)) == 0) ? 3 : 33)) + 77 + 95 : 0;
            for (int o = 16; o > 0; o >>= 1) ops += __shfl_xor(ops, o, 32);
            if (live && hl == 0) atomicAdd(&opCount[a & 63], (unsigned long long)nJobs * (unsigned long long)(ops + 69 * nEdgesP));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (q >= 0) { w.entBits[eBeg + hl] = L.nb[hl]; w.entSlot[eBeg + hl] = activeSlotOf(s, w, q); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- the same predicates with the jobs' ring places PACKED over the wave (round 4) ------------------------------------------
// k_walk_pred_star runs one job per step on ALL ring places of the point, two points per wave in lockstep: 37 of 64 lanes hold a
// place, and the move of one neighbour touches about half of them.  What it does not touch cannot decide the job: with p at its
// current position an untouched pair keeps its current angle, which is >= ptMin and <= ptMax by definition (SM.C:938-975), so
// only touched pairs can make `new < current` true (SM.C:1421-1427); with p at its proposal an untouched pair has exactly the
// angle the self test found for it.  So a job is only its TOUCHED places, and here the tasks (job, place) of both points of a
// wave are dealt densely to the 64 lanes: a lane fetches its place's record from LDS (ring faces i and i + 1, the cell's centre,
// the edge's far end), forms BOTH face vectors itself (no ring neighbour to take one from) and its cell's, and the per-job
// min / max are LDS atomics on the angles' bit patterns (positive doubles order like unsigned integers).  Measured on the 10 M-cell
// cavity mesh: 10.3 M tasks per launch against 29.3 M lane-steps of the star form.  The arithmetic per (edge, cell) pair is
// starLaneAngle's, operation for operation.
#ifndef SMGPU_PACK_WAVES
#define SMGPU_PACK_WAVES 4
#endif
#ifndef SMGPU_PACK_SEQ
#define SMGPU_PACK_SEQ 0
#endif
#ifndef SMGPU_PACK_UNROLL
#define SMGPU_PACK_UNROLL 2      // vertices of a ring face fetched together (4 / 2 / 1)
#endif
constexpr int kPackBlock = 256;
// Vertex slots of a star: 64 since round 6, 96 before (k_walk_pred_star: 128; an interior hex point has 48, the refinement interfaces
// of the castellated meshes at most 60; a larger star is left to the general kernels).  Measured before the coordinate table below, on the 10 M-cell
// cavity mesh (profiles/r4/ab_walk_pred_pack.txt): 64 slots and FOUR waves per SIMD 724 us (128 VGPRs, 36 of them spilled), 64
// slots at three waves 693 (168 VGPRs), 128 slots at three waves 697 -- the kernel is not occupancy bound.  Round 5, with the other
// kernels' occupancy steps in mind, once more: -DSMGPU_PACK_WAVES=4 -DSMGPU_PACK_VERTS=60 (40.3 KB per block, 128 VGPRs with 41 spilled) 937
// against 673 us for the launch group: four waves only help if the registers fit, and they do not.
// Round 6: FOUR waves per SIMD after all (k_fa_pred 685 -> 620 us on the 10 M-cell cavity mesh, 188 -> 173 on the 1 M-cell one, same box,
// bit-equal; profiles/r6/ab_walk_four_waves.txt).  What made the registers fit: the job loop on its own needs 126 VGPRs, and with
// the ring faces' vertices fetched two at a time instead of four (SMGPU_PACK_UNROLL) the compiler keeps it free of scratch under a
// 128-VGPR cap -- the 17 values it still spills are staging state touched once per pair of stars; and the LDS: 64 vertex slots and 16
// entries (kPackEnts) are 9.2 KB per wave, 16 waves per CU.  (Round 4's four-wave build -- 64 slots, the four-vertex fetch -- had 36
// spills inside the job loop: 724 us against 693; round 5's used 60 slots, which is not a multiple of the 32 a lane deals at a time
// (now a static_assert), and 41 spills: 937 us.)
#ifndef SMGPU_PACK_VERTS
#define SMGPU_PACK_VERTS 64
#endif
constexpr int kPackVerts = SMGPU_PACK_VERTS;
// One coordinate table per half: the vertex slots [0, kPackVerts), behind them the proposals of the point's entries
// [kPackVerts, + kStarEnts) and the point itself, current (kPackCur) and proposed (kPackProp).  A job moves the point and one
// entry's neighbour hypothetically: a vertex slot whose role says "the point" / "that neighbour" is then read from the other
// INDEX of the same table -- two 32-bit selects per vertex instead of six 64-bit ones on its coordinates (selecting a V3 by
// value cost 12 v_cndmask per vertex, a quarter of a task's vector instructions).
#ifndef SMGPU_PACK_ENTS
#define SMGPU_PACK_ENTS 16
#endif
constexpr int kPackEnts = SMGPU_PACK_ENTS;      // entries (neighbours) of a star in this kernel (<= kStarEnts; hexahedral and castellated meshes: 6)
static_assert(kPackVerts % 32 == 0 && kPackEnts <= kStarEnts, "k_walk_pred_pack: vertex slots are dealt 32 at a time");
constexpr int kPackEnt0 = kPackVerts, kPackCur = kPackVerts + kPackEnts, kPackProp = kPackCur + 1, kPackSlots = kPackProp + 1;
struct PackStar {
    int fid[kStarFaces];
    int voff[kStarFaces + 1];
    unsigned char role[kPackVerts];      // kRoleSelf: the point itself, e < kStarEnts: the neighbour of entry e, kRoleOther
    double vx[kPackSlots], vy[kPackSlots], vz[kPackSlots];
    unsigned char nb[kPackEnts];
    double pMin, pMax;                   // ptMin / ptMax of the half wave's point
    int eq[kPackEnts];                   // entry e's neighbour (point id)
    struct { int fbeg[kStarFaces]; unsigned char vface[kPackVerts]; } st;   // staging only: first entry of the face in facePts,
                                                                             // the star-local face of every vertex slot
};
struct PackPlace { double ccx, ccy, ccz; unsigned char l, lNext, xEnt, xSlot, pFirst, pad[3]; };
struct PackLds {                         // per wave: two points
    PackStar h[2];
    PackPlace place[2][32];
    unsigned touch[2][kPackEnts];        // [half][entry]: the counted places the entry's neighbour touches
    unsigned long long jmin[64], jmax[64];   // per job: bit patterns of the smallest / largest angle
    double selfAng[2][32];               // the self test's angle of every counted place
    int off[65];                         // first task of every job
    unsigned cmask[2];                   // counted places of the half
    unsigned char jcode[64];             // job t: bit 7 = half, low bits = entry, 127 = the self test
    unsigned char hflags[2], selfBad[2]; // bit 0 moved, bit 1 self test needed
};
static_assert(sizeof(PackLds) * (kPackBlock / 64) * SMGPU_PACK_WAVES <= 160 * 1024, "k_walk_pred_pack: LDS per CU");

// ---- the STATIC part of a star, kept between iterations (round 6) ----------------------------------------------------------------
// Staging a star was 228 of the kernel's 598 us on the 10 M-cell cavity mesh (a build without the job loops): five levels of
// dependent gathers through the addressing (the point's edge and face lists -> their ranges -> vertex ids, ring faces and cells ->
// coordinates), the roles of the vertex slots, the ring places with their face slots, the places every entry touches.  All of it but
// the coordinates and the neighbours' states is TOPOLOGY: the same every iteration a point is active -- and the points outside the
// good range are the same few per cent of the mesh iteration after iteration (refinement interfaces).  So the first walk a point
// takes part in leaves its record here (k_walk_star_build: the dynamic staging, dumped), and every later one (k_walk_pred_cached)
// reads the record with coalesced loads and gathers coordinates only: ONE level of dependent loads.  A pool of records with a bump
// counter; slot[p]: -1 not built yet, -2 the star does not fit the packed form (the general kernels take it), -3 the pool was full
// (k_walk_pred_pack_rest stages the star every iteration, as before).
struct StarRec {
    int vid[kPackVerts];                       // vertex slot -> point id, -1 beyond the star's slots
    unsigned char role[kPackVerts];
    unsigned char voff[kStarFaces + 4];        // first vertex slot of every face (<= kPackVerts <= 255), one behind the last
    int q[kPackEnts];                          // the entries' neighbours, -1 beyond the point's entries
    unsigned touch[kPackEnts];                 // the counted places every entry touches
    struct Place { int rc; unsigned char l, lNext, xEnt, xSlot, pFirst, counts, pad[2]; } place[32];
    unsigned cmask;                            // the counted places
    int opsPerJob;                             // packJobOps (timing passes)
    unsigned char nEnt, nFaces, totalV, nPlaces;
    int pad[4];
};
static_assert(kPackVerts <= 255 && sizeof(StarRec) % 16 == 0, "StarRec: 16-byte records");
struct StarCache { int* slot; StarRec* pool; int* count; int capacity; };      // count: [0] records handed out, [1] / [2] this iteration's needs (k_walk_count / k_walk_fill)

// position of the r-th (0-based) set bit of m (r < popcount(m))
__device__ __forceinline__ int nthSetBit(unsigned m, int r) {
    int pos = 0;
#pragma unroll
    for (int w = 16; w > 0; w >>= 1) {
        const unsigned low = m & ((1u << w) - 1u);
        const int c = __popc(low);
        if (r >= c) { r -= c; m >>= w; pos += w; }
        else m = low;
    }
    return pos;
}

// one task: the angle of place R of the half whose star is L, with p at c1 and entry ei's neighbour at c2 (ei = kRoleNoEntry: nobody)
__device__ __forceinline__ double packTaskAngle(const PackStar& L, const PackPlace& R, int selfIdx, int ei) {
    // selfIdx: where the point is in this job (kPackCur / kPackProp); ei: the entry whose neighbour sits at its proposal
    // (kRoleNoEntry: nobody); both are read through their slots of the coordinate table
    const int entIdx = kPackEnt0 + (ei < kPackEnts ? ei : 0);
    const int xs_ = ((int)R.xEnt == ei) ? entIdx : (int)R.xSlot;       // the edge's far end, possibly the moved neighbour
    const bool pFirst = R.pFirst != 0;
    const int i0 = pFirst ? selfIdx : xs_, i1 = pFirst ? xs_ : selfIdx;    // the edge's start and end (edges[2e], edges[2e + 1])
    const V3 e0 = v3(L.vx[i0], L.vy[i0], L.vz[i0]), e1 = v3(L.vx[i1], L.vy[i1], L.vz[i1]);
    const V3 cC = 0.5 * (e0 + e1);
    const V3 d = e1 - e0;
    const V3 eVec = d / mag(d);
    auto faceVec = [&](int l) -> V3 {
        V3 fc = v3(0, 0, 0);   // calcFaceCenter SM.C:1103-1130
        const int b = L.voff[l], n = L.voff[l + 1] - b;
        // the vertex in slot k with the job's substitutions: read unconditionally, selected by component -- no branch between the LDS
        // reads of a face, so that they are in flight together (the additions keep the vertex order)
        auto vert = [&](int k) -> V3 {
            const int r = L.role[k];
            const int at = (r == kRoleSelf) ? selfIdx : (r == ei) ? entIdx : k;
            return v3(L.vx[at], L.vy[at], L.vz[at]);
        };
        int i = 0;
#if SMGPU_PACK_UNROLL == 4
        for (; i + 4 <= n; i += 4) {
            const V3 q0 = vert(b + i), q1 = vert(b + i + 1), q2 = vert(b + i + 2), q3 = vert(b + i + 3);
            fc = fc + q0; fc = fc + q1; fc = fc + q2; fc = fc + q3;
        }
#elif SMGPU_PACK_UNROLL == 2
        for (; i + 2 <= n; i += 2) {
            const V3 q0 = vert(b + i), q1 = vert(b + i + 1);
            fc = fc + q0; fc = fc + q1;
        }
#endif
        for (; i < n; ++i) fc = fc + vert(b + i);
        fc = divByCount(fc, n);   // = fc / double(n), bit for bit
        const V3 cf = cC - fc;
        const double dp = dot(cf, eVec);
        const V3 pC = fc + dp * eVec;
        const V3 w = pC - cC;
        return w / mag(w);
    };
#if SMGPU_PACK_SEQ
    // one sub-computation at a time (scheduling barriers): the cell's vector, then each face's vector straight into its cosine --
    // fewer values alive at once (four waves per SIMD need <= 128 VGPRs)
    __builtin_amdgcn_sched_barrier(0);
    const V3 cc = v3(R.ccx, R.ccy, R.ccz);
    const V3 cf = cC - cc;
    const double dp = dot(cf, eVec);
    const V3 pC = cc + dp * eVec;
    const V3 w = pC - cC;
    const V3 cV = w / mag(w);
    __builtin_amdgcn_sched_barrier(0);
    const double c0 = dot(faceVec(R.l), cV);
    __builtin_amdgcn_sched_barrier(0);
    const double c1 = dot(cV, faceVec(R.lNext));
    __builtin_amdgcn_sched_barrier(0);
    return clampAcos(c0) + clampAcos(c1);   // calcEdgeCenterEdgeAngle SM.C:980-998
#else
    const V3 fv = faceVec(R.l), fn = faceVec(R.lNext);
    const V3 cc = v3(R.ccx, R.ccy, R.ccz);
    const V3 cf = cC - cc;
    const double dp = dot(cf, eVec);
    const V3 pC = cc + dp * eVec;
    const V3 w = pC - cC;
    const V3 cV = w / mag(w);
    return clampAcos(dot(fv, cV)) + clampAcos(dot(cV, fn));   // calcEdgeCenterEdgeAngle SM.C:980-998
#endif
}

// the jobs W.jcode[0 .. nJobs) of the wave (phase 0: self tests and current-position entries; phase 1: proposal-position entries):
// tasks dealt to the lanes, per-job min / max, verdicts into the halves' tables.  Called by all 64 lanes.
__device__ __forceinline__ void packRunJobs(PackLds& W, int lane, int nJobs, int phase, const Prm& prm, const State& s) {
    const unsigned long long twoPi = (unsigned long long)__double_as_longlong(2.0 * SMGPU_PI);
    // lane t = job t: its places, their number, the running offsets
    unsigned M = 0u;
    int code = 0;
    if (lane < nJobs) {
        code = W.jcode[lane];
        const int h = code >> 7, e = code & 127;
        M = (e == 127) ? W.cmask[h] : (W.touch[h][e] & W.cmask[h]);
        W.jmin[lane] = twoPi; W.jmax[lane] = 0ull;
    }
    const int c = __popc(M);
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int T = __shfl(incl, 63, 64);
    W.off[lane] = incl - c;
    if (lane == 63) W.off[64] = T;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // proposal-position jobs: the places an entry does NOT touch keep the self test's angle
    if (phase == 1) {
        const int h = lane >> 5, i = lane & 31;
        if ((W.cmask[h] >> i) & 1u) {
            const unsigned long long a = (unsigned long long)__double_as_longlong(W.selfAng[h][i]);
            for (int t = 0; t < nJobs; ++t) {
                const int cd = W.jcode[t];
                if ((cd >> 7) != h) continue;
                if ((W.touch[h][cd & 127] >> i) & 1u) continue;
                atomicMin(&W.jmin[t], a); atomicMax(&W.jmax[t], a);
            }
        }
    }
    for (int k0 = 0; k0 < T; k0 += 64) {                            // (wave-uniform)
        const int k = k0 + lane;
        if (k < T) {
            int lo = 0, hi = nJobs;                                 // the last job t with off[t] <= k (it has tasks: off[t + 1] > k)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (W.off[mid] <= k) lo = mid; else hi = mid; }
            const int t = lo, cd = W.jcode[t], h = cd >> 7, e = cd & 127;
            const unsigned Mt = (e == 127) ? W.cmask[h] : (W.touch[h][e] & W.cmask[h]);
            const int i = nthSetBit(Mt, k - W.off[t]);
            const PackStar& L = W.h[h];
            const bool isSelf = e == 127;
            const int selfIdx = (isSelf || phase == 1) ? kPackProp : kPackCur;     // where the point is in this job
            const double angle = packTaskAngle(L, W.place[h][i], selfIdx, isSelf ? (int)kRoleNoEntry : e);
            // (positive doubles order like their bit patterns.  The angle is never NaN: it is a sum of two clampAcos values, and
            // clampAcos -- the reference's std::max(-MAX, std::min(MAX, cosA)), SM.C:782-783, 992-995 -- maps a NaN cosine, e.g.
            // from a zero-length projected vector of a collapsed cell, to acos(MAX); the star form and the oracle see the same)
            const unsigned long long ab = (unsigned long long)__double_as_longlong(angle);
            atomicMin(&W.jmin[t], ab); atomicMax(&W.jmax[t], ab);
            if (isSelf) W.selfAng[h][i] = angle;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < nJobs) {                                              // the verdicts (SM.C:1391-1399, 1421-1427)
        const int h = code >> 7, e = code & 127;
        const double mn = __longlong_as_double((long long)W.jmin[lane]), mx = __longlong_as_double((long long)W.jmax[lane]);
        const double curMin = W.h[h].pMin, curMax = W.h[h].pMax;
        const bool isBad = ((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax));
        noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
        if (isBad) {
            if (e == 127) W.selfBad[h] = 1;
            else {
                const bool moved = W.hflags[h] & 1;
                unsigned char v = W.h[h].nb[e];
                v |= (phase == 1) ? 1 : (moved ? 2 : 3);             // (not moved: proposal = current position)
                W.h[h].nb[e] = v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// The algorithmic FP64 instructions of ONE job of a star, by the reference's arithmetic (SM.C:1135-1231 per (edge, cell) pair: two
// projected face-centre vectors 77 + 3 (n - 1) + (3 | 33) each for a face of n vertices, the projected cell centre 77, two clamped
// acos and their sum 95 -- sqrt = 22, division = 11 per component, acos = 40; per edge of the point 69 for the edge vector).  Called by
// the 32 lanes of a half; nv = the vertex count of the lane's ring face (counted places only).
__device__ __forceinline__ int packJobOps(bool counts, int nv, int nEdgesP) {
    int ops = counts ? 2 * (77 + 3 * (nv - 1) + (((nv & (nv - 1)) == 0) ? 3 : 33)) + 77 + 95 : 0;
    for (int o = 16; o > 0; o >>= 1) ops += __shfl_xor(ops, o, 32);
    return ops + 69 * nEdgesP;
}
// Everything behind the staging of a pair of stars (k_walk_pred_pack / k_walk_pred_cached): the job lists of both phases, the
// verdicts, the point's and the entries' bits.  Called by all 64 lanes.
__device__ __forceinline__ void packFinish(PackLds& W, PackStar& L, const State& s, const WalkView& w, const Prm& prm, int lane, int half, int hl, int a,
                                           bool live, bool moved, bool frozenBefore, bool eligible, int q, int eBeg, bool counts, int opsPerJob,
                                           unsigned long long* opCount) {
        const unsigned countedMask = (unsigned)(__ballot(counts) >> (32 * half));
        unsigned sbits = (moved ? 2u : 0u) | (frozenBefore ? 4u : 0u);
        const bool selfNeeded = live && moved && !frozenBefore;
        const unsigned elig = (unsigned)(__ballot(eligible) >> (32 * half));
        const int nEl = live ? __popc(elig) : 0, first = selfNeeded ? 1 : 0;
        const int myRank = __popc(elig & ((1u << hl) - 1u));
        // ---- phase 1: the self test (all counted places, p at its proposal) and every eligible entry with p at its CURRENT position
        // (the places it touches).  Job t of the wave: half 0's jobs, then half 1's.
        const int nJ = live ? first + nEl : 0;
        const int nJ0 = __shfl(nJ, 0, 64), nJ1 = __shfl(nJ, 32, 64);
        const int jb = half ? nJ0 : 0;
        if (hl == 0) { W.hflags[half] = (unsigned char)((moved ? 1 : 0) | (selfNeeded ? 2 : 0)); W.selfBad[half] = 0; W.cmask[half] = countedMask; }
        if (hl == 0 && selfNeeded) W.jcode[jb] = (unsigned char)((half << 7) | 127);
        if (eligible && live) W.jcode[jb + first + myRank] = (unsigned char)((half << 7) | hl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#ifndef SMGPU_PACK_ABLATE_JOBS      // (measurement build: staging only)
        packRunJobs(W, lane, nJ0 + nJ1, 0, prm, s);
#endif
        // what phase 1 decided: the self tests; then whether the point can act from its proposal at all (SM.C:1376-1399)
        if (selfNeeded && W.selfBad[half]) sbits |= 1u;
        const bool propJobs = live && moved && !(sbits & 5u) && nEl > 0;
        // ---- phase 2: the eligible entries of the points that act from their proposal, p at its PROPOSAL: the touched places are
        // evaluated, the others keep the angle the self test found for them
        const int nP = propJobs ? nEl : 0;
        const int nP0 = __shfl(nP, 0, 64), nP1 = __shfl(nP, 32, 64);
        if (nP0 + nP1 > 0) {                                        // (wave-uniform)
            const int pb = half ? nP0 : 0;
            if (propJobs && eligible) W.jcode[pb + myRank] = (unsigned char)((half << 7) | hl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#ifndef SMGPU_PACK_ABLATE_JOBS
            packRunJobs(W, lane, nP0 + nP1, 1, prm, s);
#endif
        }
        if (live && hl == 0) w.actBits[a] = (uint8_t)sbits;
        const int nJobs = nJ + nP;
        if (opCount) {
            // timing passes only: the ALGORITHMIC FP64 instructions of this point's jobs (opsPerJob: packJobOps below) times the jobs
            // that were needed.  One add per point, spread over 64 words.
            if (live && hl == 0) atomicAdd(&opCount[a & 63], (unsigned long long)nJobs * (unsigned long long)opsPerJob);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (q >= 0) { w.entBits[eBeg + hl] = L.nb[hl]; w.entSlot[eBeg + hl] = activeSlotOf(s, w, q); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
}

// TWO active points per wave as in k_walk_pred_star (the staging is the same code), the jobs packed (packRunJobs)
// memo (measurement aid, SMGPU_WALK_MEMO_STATS=1; NULL otherwise): [0] stars whose inputs have the bits they had in the point's previous
// walk, [1] stars seen, [2 + p] the hash of point p's star inputs -- everything its jobs read: the vertex slots with their roles,
// the entries' proposals and states, the point's two positions and angle bounds, the ring places' cell centres.  Would an exact
// memo of the predicates pay?  
// The block is compiled in by -DSMGPU_WALK_MEMO=1 only (a measuring build, scripts/walk_memo_stats.py): in the product kernel its
// registers cost 28 VGPR spills.
#ifndef SMGPU_WALK_MEMO
#define SMGPU_WALK_MEMO 0
#endif
__device__ __forceinline__ unsigned long long memoMix(unsigned long long h, unsigned long long v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 33);
}
// DUMP: build the records of the active points that have none yet (k_walk_star_build) instead of evaluating anything
template <bool DUMP>
__device__ __forceinline__ void packKernelBody(const MeshView& m, const State& s, const Prm& prm, const WalkView& w, int nA, int nE, unsigned long long* opCount,
                                               unsigned long long* memo, const StarCache& sc) {
    if (s.acc->stop) return;
    if (DUMP && sc.count[1] == 0) return;                 // every active point of this iteration has its record (k_walk_fill looked)
    if (!DUMP && sc.slot && sc.count[2] == 0) return;     // k_walk_pred_pack_rest: nobody was turned away by a full pool
    if (nA < 0) { nA = w.header[0]; nE = w.header[1]; }
    __shared__ PackLds plds[kPackBlock / 64];
    PackLds& W = plds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    PackStar& L = W.h[half];
    const int groups = gridDim.x * (kPackBlock / 32);
    // the loop is wave-uniform (half 0 decides; half 1 of the last round may be without a point: live = false)
    for (int a0 = (blockIdx.x * (kPackBlock / 64) + (threadIdx.x >> 6)) * 2; a0 < nA; a0 += groups) {
        const int a = a0 + half;
        bool live = a < nA;
        if (a == 0 && hl == 0) w.actEntOff[nA] = nE;
        const int p = live ? w.actIds[a] : 0;
        if (DUMP) {      // (wave-uniform) only the points without a record
            live = live && sc.slot[p] == -1;
            if (!__any(live)) continue;
        } else if (sc.slot) {      // behind k_walk_pred_cached: only the points the full pool could not take
            live = live && sc.slot[p] == -3;
            if (!__any(live)) continue;
        }
        bool moved;
        const bool frozenBefore = s.frozen[p] != 0;
        {   // the point itself goes to LDS (every job reads it from there)
            const V3 cur = ldv(s.ptsCur, p);
            const V3 np = ldv(s.prop, p);
            moved = (np != cur);
            if (hl == 0) {
                L.vx[kPackCur] = cur.x; L.vy[kPackCur] = cur.y; L.vz[kPackCur] = cur.z;
                L.vx[kPackProp] = np.x; L.vy[kPackProp] = np.y; L.vz[kPackProp] = np.z;
                L.pMin = s.ptMin[p]; L.pMax = s.ptMax[p];
            }
        }
        const int eBeg = live ? w.actEntOff[a] : 0, eEnd = live ? ((a + 1 < nA) ? w.actEntOff[a + 1] : nE) : 0, nEnt = eEnd - eBeg;
        // the point's edges with the lengths of their face rings, its faces with their vertex counts
        const int eb = m.ppOff[p], nEdgesP = live ? m.ppOff[p + 1] - eb : 0;
        const int fb = m.pfOff[p], nF = live ? m.pfOff[p + 1] - fb : 0;
        const int myE = (hl < nEdgesP) ? m.peEdge[eb + hl] : -1;
        const int myNf = (myE >= 0) ? m.efOff[myE + 1] - m.efOff[myE] : 0;
        const bool myRingBad = (myE >= 0) && !m.edgeRingOk[myE];
        const int myF = (hl < nF) ? m.pfFace[fb + hl] : -1;
        const int myFb = (myF >= 0) ? m.faceOff[myF] : 0;
        const int myV = (myF >= 0) ? m.faceOff[myF + 1] - myFb : 0;
        int inclN = myNf, inclV = myV;
        for (int o = 1; o < 32; o <<= 1) {
            const int tn = __shfl_up(inclN, o, 32), tv = __shfl_up(inclV, o, 32);
            if (hl >= o) { inclN += tn; inclV += tv; }
        }
        const int totalLanes = __shfl(inclN, 31, 32), totalV = __shfl(inclV, 31, 32);
        const unsigned ringBad = (unsigned)(__ballot(myRingBad) >> (32 * half));
        // (entry j = the neighbour across edge j: pointPoints and pointEdges share their offsets, so nEnt == nEdgesP)
        const bool fits = nEdgesP <= 32 && nF <= kStarFaces && totalLanes <= 32 && totalV <= kPackVerts && nEnt < kPackEnts && nEnt == nEdgesP && ringBad == 0u;   // (< : self + entries <= 32 jobs per half)
        if (!DUMP && live && !fits && hl == 0) { w.actBits[a] = kStarLeft; atomicAdd(&w.header[2], 1); }   // left to k_walk_pred_self / k_walk_pred (onlyLeft)
        if (DUMP && live && !fits && hl == 0) sc.slot[p] = -2;
        live = live && fits;
        // the entries of the point: lane i holds entry i (its neighbour, whether that one is free and moving)
        const int q = (live && hl < nEnt) ? w.entNbr[eBeg + hl] : -1;
        // Stage the star.  The loads are arranged in LEVELS of independent requests (what bounds this kernel besides FP64 issue
        // is the chain of dependent gathers, not their number): the edge lanes fetch their edge's cell range and end points
        // here, next to the faces' vertex ranges (the ring places below get them by shuffle instead of loading them one level
        // later), and the vertices are fetched by SLOT (four slots per lane: two round trips for the whole star) instead of by
        // a loop over each face's vertices (two dependent round trips per vertex).
        const int myCb = (myE >= 0) ? m.ecOff[myE] : 0, myNc = (myE >= 0) ? m.ecOff[myE + 1] - myCb : 0;
        const int myEfb = (myE >= 0) ? m.efOff[myE] : 0;
        const int myE0 = (myE >= 0) ? m.edges[2 * myE] : -1;
        if (live && hl < nF) {
            const int o = inclV - myV;
            L.fid[hl] = myF;
            L.voff[hl] = o;
            L.st.fbeg[hl] = myFb;
            for (int v = 0; v < myV; ++v) L.st.vface[o + v] = (unsigned char)hl;
        }
        if (live && hl == 0) L.voff[nF] = totalV;
        if (live && hl < nEnt) L.eq[hl] = q;
        if (hl < kPackEnts) W.touch[half][hl] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // this lane's ring place
        StarLane P;
        P.valid = live && hl < totalLanes;
        P.hasCell = false; P.xEnt = 0; P.xSlot = 0; P.pFirst = true; P.cc = v3(0, 0, 0); P.l = 0; P.nextLane = hl;
        int ringFaceId = -1, ringCellAt = -1;
        {
            int first = 0, nfj = 0, cb = 0, nc = 0, efb = 0, e0I = -1, jOf = 0;
            bool found = false;
            for (int j = 0; j < nEdgesP; ++j) {
                const int hi = __shfl(inclN, j, 32), nj = __shfl(myNf, j, 32);
                const int cbj = __shfl(myCb, j, 32), ncj = __shfl(myNc, j, 32), efbj = __shfl(myEfb, j, 32);
                const int e0j = __shfl(myE0, j, 32);
                if (!found && hl < hi) { found = true; first = hi - nj; nfj = nj; cb = cbj; nc = ncj; efb = efbj; e0I = e0j; jOf = j; }
            }
            if (P.valid) {
                const int i = hl - first;
                P.hasCell = i < nc;
                P.nextLane = (i + 1 < nfj) ? hl + 1 : first;          // closed ring: the last cell ends at face 0
                P.pFirst = (e0I == p);
                P.xEnt = jOf;
                ringFaceId = efb + i;
                if (P.hasCell) ringCellAt = cb + i;
            }
        }
        // level: vertex ids by slot, the ring's face and cell, the entry neighbour's two positions
        int vg[kPackVerts / 32];
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) {
            const int k = hl + 32 * u;
            vg[u] = -1;
            if (live && k < totalV) { const int l = L.st.vface[k]; vg[u] = m.facePts[L.st.fbeg[l] + (k - L.voff[l])]; }
        }
        const int rf = (ringFaceId >= 0) ? m.ringFace[ringFaceId] : -1;
        const int rc = (ringCellAt >= 0) ? m.ringCell[ringCellAt] : -1;
        V3 nq = v3(0, 0, 0);
        bool eligible = false;
        unsigned char nb0 = 0;
        if (q >= 0) {
            nq = ldv(s.prop, q);
            const bool qFrozen = s.frozen[q] != 0;
            nb0 = qFrozen ? 8 : 0;
            eligible = !qFrozen && nq != ldv(s.ptsCur, q);   // SM.C:1411-1414
            if (eligible) nb0 |= 4;
        }
        // level: the coordinates; the role of every vertex slot (the point itself / entry e's neighbour / anybody else)
        V3 vc[kPackVerts / 32];
        unsigned char role[kPackVerts / 32];
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) {
            vc[u] = (vg[u] >= 0) ? ldv(s.ptsCur, vg[u]) : v3(0, 0, 0);
            role[u] = (vg[u] == p) ? kRoleSelf : kRoleOther;
        }
        if (rc >= 0) P.cc = ldv(s.cellCtr, rc);                      // mesh.C()[cellI] of the CURRENT mesh, SM.C:1218
        for (int e = 0; e < nEnt; ++e) {
            const int qe = L.eq[e];
#pragma unroll
            for (int u = 0; u < kPackVerts / 32; ++u) if (vg[u] == qe) role[u] = (unsigned char)e;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) {
            const int k = hl + 32 * u;
            if (vg[u] >= 0) { L.role[k] = role[u]; L.vx[k] = vc[u].x; L.vy[k] = vc[u].y; L.vz[k] = vc[u].z; }
        }
        if (q >= 0) { L.nb[hl] = nb0; L.vx[kPackEnt0 + hl] = nq.x; L.vy[kPackEnt0 + hl] = nq.y; L.vz[kPackEnt0 + hl] = nq.z; }
        if (P.valid) for (int l = 0; l < nF; ++l) if (L.fid[l] == rf) P.l = l;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // the slot of the edge's other end point in this lane's ring face; the place's record for the packed steps; the places
        // every entry TOUCHES: an angle depends on the neighbour q_e through the edge (its own entry) and through the vertices of
        // the place's two ring faces (SM.C:1155-1200) -- nothing else
        const int lNext = __shfl(P.l, P.nextLane, 32);
        const bool counts = P.valid && P.hasCell;
        if (P.valid) for (int i = L.voff[P.l]; i < L.voff[P.l + 1]; ++i) if (L.role[i] == P.xEnt) P.xSlot = i;
        if (counts) {
            atomicOr(&W.touch[half][P.xEnt], 1u << hl);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int l = side ? lNext : P.l;
                for (int i = L.voff[l]; i < L.voff[l + 1]; ++i) {
                    const unsigned r = L.role[i];
                    if (r < (unsigned)kPackEnts) atomicOr(&W.touch[half][r], 1u << hl);
                }
            }
            PackPlace& R = W.place[half][hl];
            R.ccx = P.cc.x; R.ccy = P.cc.y; R.ccz = P.cc.z;
            R.l = (unsigned char)P.l; R.lNext = (unsigned char)lNext; R.xEnt = (unsigned char)P.xEnt; R.xSlot = (unsigned char)P.xSlot;
            R.pFirst = P.pFirst ? 1 : 0;
        }
        if (DUMP) {      // the static part of both stars into their records
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const unsigned cm = (unsigned)(__ballot(counts) >> (32 * half));
            const int ops = packJobOps(counts, counts ? L.voff[P.l + 1] - L.voff[P.l] : 0, nEdgesP);
            int slot = -3;
            if (live && hl == 0) { slot = atomicAdd(sc.count, 1); if (slot >= sc.capacity) { slot = -3; sc.count[2] = 1; } }      // (-3: the pool is full -- k_walk_pred_pack stages the star every time)
            slot = __shfl(slot, 0, 32);
            if (live && slot >= 0) {
                StarRec& R = sc.pool[slot];
#pragma unroll
                for (int u = 0; u < kPackVerts / 32; ++u) { const int k = hl + 32 * u; R.vid[k] = vg[u]; R.role[k] = role[u]; }
                R.voff[hl] = (unsigned char)((hl <= nF) ? L.voff[hl] : 0);
                if (hl < 4) R.voff[32 + hl] = (unsigned char)((hl == 0 && nF == 32) ? L.voff[32] : 0);
                if (hl < kPackEnts) { R.q[hl] = q; R.touch[hl] = W.touch[half][hl]; }
                StarRec::Place pl;
                pl.rc = rc; pl.l = (unsigned char)P.l; pl.lNext = (unsigned char)lNext; pl.xEnt = (unsigned char)P.xEnt; pl.xSlot = (unsigned char)P.xSlot;
                pl.pFirst = P.pFirst ? 1 : 0; pl.counts = counts ? 1 : 0; pl.pad[0] = pl.pad[1] = 0;
                R.place[hl] = pl;
                if (hl == 0) {
                    R.cmask = cm; R.opsPerJob = ops; R.nEnt = (unsigned char)nEnt; R.nFaces = (unsigned char)nF; R.totalV = (unsigned char)totalV;
                    R.nPlaces = (unsigned char)totalLanes; R.pad[0] = R.pad[1] = R.pad[2] = R.pad[3] = 0;
                }
            }
            if (live && hl == 0) sc.slot[p] = slot;
            continue;
        }
        if (SMGPU_WALK_MEMO && memo) {      // (wave-uniform) the hash of the star's inputs against the one of the point's previous walk
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            unsigned long long hsh = 0ull;
            if (live) {
                for (int k = hl; k < kPackSlots; k += 32) {
                    const bool used = k < totalV || (k >= kPackEnt0 && k < kPackEnt0 + nEnt) || k >= kPackCur;
                    if (!used) continue;
                    unsigned long long x = memoMix(0x1234567ull + (unsigned long long)k, (unsigned long long)__double_as_longlong(L.vx[k]));
                    x = memoMix(x, (unsigned long long)__double_as_longlong(L.vy[k]));
                    x = memoMix(x, (unsigned long long)__double_as_longlong(L.vz[k]));
                    if (k < totalV) x = memoMix(x, (unsigned long long)L.role[k]);
                    hsh ^= x;
                }
                if (hl < nEnt) hsh ^= memoMix(0x777ull + hl, (unsigned long long)nb0 | ((unsigned long long)(unsigned)q << 8));
                if (counts) {
                    unsigned long long x = memoMix(0x999ull + hl, (unsigned long long)__double_as_longlong(P.cc.x));
                    x = memoMix(x, (unsigned long long)__double_as_longlong(P.cc.y));
                    x = memoMix(x, (unsigned long long)__double_as_longlong(P.cc.z));
                    x = memoMix(x, ((unsigned long long)(unsigned)P.l << 24) | ((unsigned long long)(unsigned)lNext << 16) | ((unsigned long long)(unsigned)P.xEnt << 8) | (unsigned long long)(unsigned)P.xSlot);
                    hsh ^= x;
                }
                if (hl == 0) {
                    hsh ^= memoMix(0xabcull, (unsigned long long)__double_as_longlong(L.pMin));
                    hsh ^= memoMix(0xdefull, (unsigned long long)__double_as_longlong(L.pMax));
                    hsh ^= memoMix(0x555ull, (unsigned long long)(frozenBefore ? 1 : 0));
                }
            }
            for (int o = 16; o > 0; o >>= 1) hsh ^= __shfl_xor(hsh, o, 32);
            if (live && hl == 0) {
                if (memo[2 + (size_t)p] == hsh) atomicAdd(&memo[0], 1ull);
                atomicAdd(&memo[1], 1ull);
                memo[2 + (size_t)p] = hsh;
            }
        }
        const int opsPerJob = opCount ? packJobOps(counts, counts ? L.voff[P.l + 1] - L.voff[P.l] : 0, nEdgesP) : 0;
        packFinish(W, L, s, w, prm, lane, half, hl, a, live, moved, frozenBefore, eligible, q, eBeg, counts, opsPerJob, opCount);
    }
}

__global__ void __launch_bounds__(kPackBlock, SMGPU_PACK_WAVES) k_walk_pred_pack(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, unsigned long long* opCount,
                                                                                 unsigned long long* memo = nullptr) {
    packKernelBody<false>(m, s, prm, w, nA, nE, opCount, memo, StarCache{nullptr, nullptr, nullptr, 0});
}
// ... for the points whose records found no room in the pool (StarCache::slot == -3), behind k_walk_pred_cached
__global__ void __launch_bounds__(kPackBlock, SMGPU_PACK_WAVES) k_walk_pred_pack_rest(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, unsigned long long* opCount,
                                                                                      StarCache sc) {
    packKernelBody<false>(m, s, prm, w, nA, nE, opCount, nullptr, sc);
}
__global__ void __launch_bounds__(kPackBlock, SMGPU_PACK_WAVES) k_walk_star_build(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, StarCache sc) {
    packKernelBody<true>(m, s, prm, w, nA, nE, nullptr, nullptr, sc);
}

// The same predicates from the stars' records (StarRec): the static tables arrive with coalesced loads, the only dependent level
// left is the coordinates -- the star's vertices, its ring cells' centres, the entries' proposals -- and the neighbours' states.
// Points without a record (slot < 0: the star does not fit, or the pool is full) are left to the general kernels like the stars
// k_walk_pred_pack cannot take.  What the jobs read is, array for array, what k_walk_pred_pack's staging leaves in LDS.
__global__ void __launch_bounds__(kPackBlock, SMGPU_PACK_WAVES) k_walk_pred_cached(MeshView m, State s, Prm prm, WalkView w, int nA, int nE, unsigned long long* opCount,
                                                                                   StarCache sc) {
    if (s.acc->stop) return;
    if (nA < 0) { nA = w.header[0]; nE = w.header[1]; }
    __shared__ PackLds plds[kPackBlock / 64];
    PackLds& W = plds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    PackStar& L = W.h[half];
    const int groups = gridDim.x * (kPackBlock / 32);
    for (int a0 = (blockIdx.x * (kPackBlock / 64) + (threadIdx.x >> 6)) * 2; a0 < nA; a0 += groups) {
        const int a = a0 + half;
        bool live = a < nA;
        if (a == 0 && hl == 0) w.actEntOff[nA] = nE;
        const int p = live ? w.actIds[a] : 0;
        const int slot = live ? sc.slot[p] : -1;
        const int eBeg = live ? w.actEntOff[a] : 0, eEnd = live ? ((a + 1 < nA) ? w.actEntOff[a + 1] : nE) : 0, nEnt = eEnd - eBeg;
        const StarRec& R = sc.pool[slot >= 0 ? slot : 0];
        // round 1: the record (static) and the point itself
        const bool frozenBefore = s.frozen[p] != 0;
        const V3 cur = ldv(s.ptsCur, p);
        const V3 np = ldv(s.prop, p);
        const double pMin = s.ptMin[p], pMax = s.ptMax[p];
        int vg[kPackVerts / 32];
        unsigned char role[kPackVerts / 32];
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) { vg[u] = R.vid[hl + 32 * u]; role[u] = R.role[hl + 32 * u]; }
        const unsigned voffWord = reinterpret_cast<const unsigned*>(R.voff)[hl < (kStarFaces + 4) / 4 ? hl : 0];
        const int qRec = R.q[hl < kPackEnts ? hl : 0];
        const unsigned touchRec = R.touch[hl < kPackEnts ? hl : 0];
        const StarRec::Place pl = R.place[hl];
        const int recEnt = R.nEnt, opsPerJob = R.opsPerJob;
        const bool fits = slot >= 0 && recEnt == nEnt;
        // (slot -3: k_walk_pred_pack_rest takes the point; everything else without a usable record is left to k_walk_pred_self / k_walk_pred)
        if (live && !fits && slot != -3 && hl == 0) { w.actBits[a] = kStarLeft; atomicAdd(&w.header[2], 1); }
        live = live && fits;
        const bool moved = (np != cur);
        const int q = (live && hl < nEnt) ? qRec : -1;
        const bool counts = live && pl.counts != 0;
        // round 2: the coordinates and the neighbours' states
        V3 vc[kPackVerts / 32];
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) { if (!live) vg[u] = -1; vc[u] = (vg[u] >= 0) ? ldv(s.ptsCur, vg[u]) : v3(0, 0, 0); }
        V3 cc = v3(0, 0, 0);
        if (counts && pl.rc >= 0) cc = ldv(s.cellCtr, pl.rc);        // mesh.C()[cellI] of the CURRENT mesh, SM.C:1218
        V3 nq = v3(0, 0, 0);
        bool eligible = false;
        unsigned char nb0 = 0;
        if (q >= 0) {
            nq = ldv(s.prop, q);
            const bool qFrozen = s.frozen[q] != 0;
            nb0 = qFrozen ? 8 : 0;
            eligible = !qFrozen && nq != ldv(s.ptsCur, q);   // SM.C:1411-1414
            if (eligible) nb0 |= 4;
        }
        // into LDS, as k_walk_pred_pack's staging leaves it
        if (hl == 0) {
            L.vx[kPackCur] = cur.x; L.vy[kPackCur] = cur.y; L.vz[kPackCur] = cur.z;
            L.vx[kPackProp] = np.x; L.vy[kPackProp] = np.y; L.vz[kPackProp] = np.z;
            L.pMin = pMin; L.pMax = pMax;
        }
        if (live && hl < (kStarFaces + 4) / 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) if (4 * hl + b <= kStarFaces) L.voff[4 * hl + b] = (int)((voffWord >> (8 * b)) & 0xffu);
        }
#pragma unroll
        for (int u = 0; u < kPackVerts / 32; ++u) {
            const int k = hl + 32 * u;
            if (vg[u] >= 0) { L.role[k] = role[u]; L.vx[k] = vc[u].x; L.vy[k] = vc[u].y; L.vz[k] = vc[u].z; }
        }
        if (q >= 0) { L.nb[hl] = nb0; L.vx[kPackEnt0 + hl] = nq.x; L.vy[kPackEnt0 + hl] = nq.y; L.vz[kPackEnt0 + hl] = nq.z; }
        if (hl < kPackEnts) W.touch[half][hl] = live ? touchRec : 0u;
        if (counts) {
            PackPlace& Rp = W.place[half][hl];
            Rp.ccx = cc.x; Rp.ccy = cc.y; Rp.ccz = cc.z;
            Rp.l = pl.l; Rp.lNext = pl.lNext; Rp.xEnt = pl.xEnt; Rp.xSlot = pl.xSlot; Rp.pFirst = pl.pFirst;
        }
        packFinish(W, L, s, w, prm, lane, half, hl, a, live, moved, frozenBefore, eligible, q, eBeg, counts, opsPerJob, opCount);
    }
}


// ---- second compaction (over active slots) ------------------------------------------------------------------
__device__ __forceinline__ bool entryActs(uint8_t nb) { return (nb & 4) && (nb & 3); }   // moving neighbour, hurt in some state

// The counts come from the device header (no host read-back); the launch covers every possible slot (kRelPer consecutive slots
// per thread) and the workgroups beyond the count leave at once.  No scan launch between count and fill: k_rel_fill sums the
// per-workgroup counts itself (chunkPrefix2) -- all of them, because the item positions count from the END of the sequence --
// and leaves {nRelevant, nBadEntries} in w.header2.
constexpr int kRelPer = 4;
constexpr int kRelChunk = kBlock * kRelPer;
inline int relGrid(int64_t n) { return (int)std::max<int64_t>(1, (n + kRelChunk - 1) / kRelChunk); }
__device__ __forceinline__ void relOf(const WalkView& w, int a, int& r, int& b) {
    b = 0;
    for (int k = w.actEntOff[a]; k < w.actEntOff[a + 1]; ++k) b += entryActs(w.entBits[k]) ? 1 : 0;
    const uint8_t sb = w.actBits[a] & 0x7f;
    r = (b > 0 || ((sb & 1) && (sb & 2))) ? 1 : 0;
    if (!r) b = 0;
}
__global__ void __launch_bounds__(kBlock) k_rel_count(WalkView w) {
    const int nA = w.header[0];
    if ((int)(blockIdx.x * kRelChunk) >= nA) return;
    const int a0 = blockIdx.x * kRelChunk + threadIdx.x * kRelPer;
    int r = 0, b = 0;
    for (int i = 0; i < kRelPer; ++i)
        if (a0 + i < nA) { int ri, bi; relOf(w, a0 + i, ri, bi); r += ri; b += bi; }
    chunkReduce2(r, b);
    if (threadIdx.x == 0) { w.blkA[blockIdx.x] = r; w.blkE[blockIdx.x] = b; }
}

// fx.T != NULL (device replay): an entry item keeps its owner's slot in `hpos` (k_rel_link leaves it there: nothing is
// pushed in that mode) and the barrier words of k_walk_fix are reset
__global__ void __launch_bounds__(kBlock) k_rel_fill(WalkView w, FixView fx) {
    if (fx.T && blockIdx.x == 0 && threadIdx.x == 0) { fx.bar[0] = fx.bar[1] = fx.bar[2] = fx.bar[3] = 0u; fx.flags[0] = fx.flags[1] = fx.flags[2] = fx.flags[3] = 0; }
    const int nA = w.header[0];
    if (nA <= 0) { if (blockIdx.x == 0 && threadIdx.x == 0) { w.header2[0] = 0; w.header2[1] = 0; } return; }
    if ((int)(blockIdx.x * kRelChunk) >= nA) return;
    const int a0 = blockIdx.x * kRelChunk + threadIdx.x * kRelPer;
    int rr[kRelPer], bb[kRelPer];
    int r = 0, b = 0;
#pragma unroll
    for (int i = 0; i < kRelPer; ++i) {
        rr[i] = bb[i] = 0;
        if (a0 + i < nA) { relOf(w, a0 + i, rr[i], bb[i]); r += rr[i]; b += bb[i]; }
    }
    int xr, xb, pr, pb, nR, nB;
    chunkScan2(r, b, xr, xb);
    chunkPrefix2(w.blkA, w.blkE, (int)blockIdx.x, (nA + kRelChunk - 1) / kRelChunk, true, pr, pb, nR, nB);
    if (blockIdx.x == 0 && threadIdx.x == 0) { w.header2[0] = nR; w.header2[1] = nB; }
    int slot = pr + xr, eoEnd = pb + xb;
#pragma unroll
    for (int i = 0; i < kRelPer; ++i) {
        const int a = a0 + i;
        if (a >= nA) break;
        w.relSlot[a] = rr[i] ? slot : -1;
        if (!rr[i]) continue;
        eoEnd += bb[i];                                    // entries of slots 0..slot
        // visiting order: slot nR-1 first; every earlier-visited point contributes its header and its entries
        int pos = (nR - 1 - slot) + (nB - eoEnd);
        w.hdrPos[slot] = pos;
        const int hdr = pos++;
        uint8_t anyOld = 0;
        for (int k = w.actEntOff[a]; k < w.actEntOff[a + 1]; ++k) {
            const uint8_t nb = w.entBits[k];
            if (!entryActs(nb)) continue;
            w.items[pos] = WalkItem{w.entSlot[k], w.entNbr[k], nb, slot};   // active slot for now, see k_rel_link
            anyOld |= nb & 2;
            ++pos;
        }
        // bit 3: a re-visit of this point after a neighbour froze it (held at its current position, SM.C:1431)
        // can freeze somebody; without it the replay does not need to come back
        const uint8_t rb = (w.actBits[a] & 0x7f) | (anyOld ? 8 : 0);
        w.relBits[slot] = rb;
        const bool moved = rb & 2, selfBad = rb & 1;
        w.items[hdr] = WalkItem{slot, w.actIds[a], ((moved && selfBad) ? 3u : 0u) | 32u | ((moved && !selfBad) ? 64u : 0u) | 128u, hdr};
        ++slot;
    }
}

__global__ void __launch_bounds__(kBlock) k_rel_link(WalkView w, int nItems, int nR, int keepOwner) {
    if (nItems < 0) { nR = w.header2[0]; nItems = nR + w.header2[1]; }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nItems; i += gridDim.x * kBlock) {
        WalkItem it = w.items[i];
        if (it.bits & 0x80u) continue;
        // a neighbour that is not active, or active but unable to act, is a sink: the replay only honours its
        // "frozen before the walk" bit (two constant pseudo slots behind the real ones)
        const unsigned nb = it.bits;
        const int rs = (it.slot >= 0) ? w.relSlot[it.slot] : -1;
        const int owner = it.hpos;
        unsigned bits = nb & 3u;
        it.hpos = 0;
        if (rs >= 0) {
            it.slot = rs;
            it.hpos = w.hdrPos[rs];
            bits |= 32u;
            if (w.relBits[rs] & 8) bits |= 16u;
        } else {
            it.slot = nR + ((nb >> 3) & 1u);
        }
        if (keepOwner) it.hpos = owner;
        it.bits = bits;
        w.items[i] = it;
    }
}

// ---- device replay: the walk as a causal fixed point (see the head of this file) ---------------------------------------
constexpr int kFixBlock = 1024;
constexpr int kNever = 0x7fffffff;
__device__ __forceinline__ int ldAgent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stAgent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier of the persistent launch.  Everything the workgroups hand each other (T, act, flags, the items' last-sent
// words) is written and read with agent-scope atomics / sc1 accesses, which are served by L2, so the barrier only has to
// make sure that every wave's operations have completed before its workgroup arrives (MI355X_MICROARCH.md, "Valid forms":
// agent atomics on both sides).  A barrier that cannot complete (workgroups that never became resident: several engines or
// a foreign process on the device) must not hang the stream: the first workgroup whose wait exceeds the wall-clock limit
// raises acc->err AND an abort word (flags[3]) that every spin loop tests, so all workgroups leave within microseconds of
// each other; k_walk_fix returns as soon as a barrier reports it.  Returns false when the launch is being abandoned.
constexpr unsigned long long kFixTimeoutTicks = 200000000ull;   // s_memrealtime runs at 100 MHz: two seconds
// One 64-bit atomic per workgroup and barrier: the low word counts arrivals, the high word the workgroups that report a change
// (the vote rides in the arrival: a separate flag word cost a second device-scope atomic round trip, ~2 us of the ~7 us a
// barrier took).  Two counters are used alternately: a workgroup that runs ahead into the next barrier adds to the other
// counter and cannot come back to this one before everybody has read it.  Returns false when the launch is being abandoned;
// *changed = some workgroup reported a change in THIS barrier.
// (plain words, no arrays: a dynamically indexed private array is promoted to LDS, 4 bytes x 1024 threads per element)
struct FixSync { unsigned tgt0, tgt1, seen0, seen1, n, nAny; };
// "does any thread of the workgroup say yes" through three rotating LDS words (the library's __syncthreads_or carries its own
// 20 KB of LDS per kernel): word u is written before the barrier of use u and read after it; thread 0 clears it during use
// u + 2, i.e. behind barrier u + 1, which no thread passes before its read of use u, and in front of barrier u + 2, behind
// which the word is written again (use u + 3).
__device__ __forceinline__ bool fixAny(int* anyWords, FixSync& fs, bool mine) {
    const unsigned u = fs.nAny % 3u;
    ++fs.nAny;
    if (threadIdx.x == 0) anyWords[(u + 1u) % 3u] = 0;
    if (mine) anyWords[u] = 1;
    __syncthreads();
    return anyWords[u] != 0;
}
__device__ __forceinline__ bool fixSync(FixView fx, FixSync& fs, int* anyWords, bool mine, Accum* acc, bool* changed) {
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(fx.bar);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool any = fixAny(anyWords, fs, mine);
    __shared__ int shAbort, shChanged;
    if (threadIdx.x == 0) {
        const unsigned c = fs.n & 1u;
        if (c) fs.tgt1 += gridDim.x; else fs.tgt0 += gridDim.x;
        const unsigned tgt = c ? fs.tgt1 : fs.tgt0;
        int ab = 0;
        __hip_atomic_fetch_add(&ctr[c], 1ull | (any ? (1ull << 32) : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        unsigned long long v = __hip_atomic_load(&ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int)((unsigned)v - tgt) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u) {
                if (ldAgent(&fx.flags[3]) != 0) { ab = 1; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > kFixTimeoutTicks) {
                    acc->err = 3;
                    stAgent(&fx.flags[3], 1);
                    ab = 1;
                    break;
                }
            }
            v = __hip_atomic_load(&ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!ab && ldAgent(&fx.flags[3]) != 0) ab = 1;
        const unsigned high = (unsigned)(v >> 32);
        shChanged = (high != (c ? fs.seen1 : fs.seen0)) ? 1 : 0;
        if (c) fs.seen1 = high; else fs.seen0 = high;
        shAbort = ab;
    }
    ++fs.n;
    __syncthreads();
    if (changed) *changed = shChanged != 0;
    return shAbort == 0;
}

constexpr int kFixPer = 4, kFixLoc = 2048;   // (small on purpose: several engines can share a device, and every workgroup of
                                              // every persistent launch has to be resident at once)   // items per thread in registers; owners' T values per workgroup in LDS (2 x 24 KB)
// (two workgroups per CU = 8 waves per SIMD = at most 64 VGPRs: several engines on one device -- ranks sharing a GPU in tests,
// the sub-domains of LocalMultiSmoother -- need their persistent launches resident side by side)
__global__ void __launch_bounds__(kFixBlock, 8) k_walk_fix(WalkView w, FixView fx, State s, int maxSweeps, int useLocalArg) {
    if (s.acc->stop) return;
    const int nR = w.header2[0], nItems = nR + w.header2[1];
    if (nR <= 0) return;
    const int gtid = blockIdx.x * kFixBlock + threadIdx.x, gstride = gridDim.x * kFixBlock;
    FixSync fs = {0u, 0u, 0u, 0u, 0u, 0u};
    __shared__ int anyWords[3];
    if (threadIdx.x < 3) anyWords[threadIdx.x] = 0;
    __syncthreads();
    bool ok = true;
    // First guess of the set A (points that act from their proposal): what the same point did in the previous iteration's walk.
    // Any start converges to the one fixed point (every round gets the earliest wrong decision and everything before it
    // right); consecutive smoothing iterations differ little, so the guess is mostly right and the loop below ends after one
    // or two rounds instead of three (fx.actPrev == NULL: the empty set, round 2's start).
    for (int x = gtid; x < nR; x += gstride) {
        int g = 0;
        if (fx.actPrev && (w.relBits[x] & 7u) == 2u) g = fx.actPrev[w.items[w.hdrPos[x]].id] ? 1 : 0;
        stAgent(&fx.act[x], g);
    }
    // this workgroup's slab of the item sequence; its propagating items (current-state entries with a real target whose owner
    // moves and was not frozen before the walk) in registers, the slot range of its owners for the LDS copy of T
    const int chunk = (nItems + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(nItems, i0 + chunk);
    __shared__ int Tloc[kFixLoc];
    __shared__ unsigned dirty[kFixLoc / 32];   // slots the slab lowered itself since the last reload
    __shared__ int shLo, shHi;
    int myO[kFixPer], myT[kFixPer], myLast[kFixPer];
    if (threadIdx.x == 0) { shLo = 0x7fffffff; shHi = -1; }
    __syncthreads();
    {
        int lo = 0x7fffffff, hi = -1;
#pragma unroll
        for (int k = 0; k < kFixPer; ++k) {
            myO[k] = -1; myT[k] = 0; myLast[k] = kNever;
            const int i = i0 + threadIdx.x + k * kFixBlock;
            if (i < i1) {
                const WalkItem it = w.items[i];
                if (!(it.bits & 0x80u) && (it.bits & 34u) == 34u) {
                    const unsigned rbo = w.relBits[it.hpos];
                    if (!(rbo & 4u) && (rbo & 2u)) { myO[k] = it.hpos; myT[k] = it.slot; lo = min(lo, it.hpos); hi = max(hi, it.hpos); }
                }
            }
        }
        if (hi >= 0) { atomicMin(&shLo, lo); atomicMax(&shHi, hi); }
    }
    __syncthreads();
    const int oLo = shLo, oHi = shHi, oN = (oHi >= oLo) ? oHi - oLo + 1 : 0;
    // (wave-uniform per workgroup; a slab too long for the registers or too wide for the LDS copy keeps the global form)
    const bool useLocal = useLocalArg && chunk <= kFixPer * kFixBlock && oN <= kFixLoc;
    const unsigned long long tStart = __builtin_amdgcn_s_memrealtime();
    unsigned long long tSweeps = 0;
    int nVotes = 0, nOuter = 0;
    for (int outer = 0;; ++outer) {
        ++nOuter;
        // the seeds that do not depend on anybody else
        for (int x = gtid; x < nR; x += gstride) {
            const unsigned rb = w.relBits[x];   // bit0 own move deteriorates, bit1 moved, bit2 frozen before the walk
            stAgent(&fx.T[x], (rb & 4u) ? -1 : (((rb & 3u) == 3u) ? w.hdrPos[x] : kNever));
        }
        if (!fixSync(fx, fs, anyWords, false, s.acc, nullptr)) return;
        // entries that fire at their owner's first visit: proposal-state entries of the owners in A; current-state entries
        // of owners that never move or were frozen before the walk (they are never re-visited)
        for (int i = gtid; i < nItems; i += gstride) {
            const WalkItem it = w.items[i];
            if ((it.bits & 0x80u) || !(it.bits & 32u)) continue;          // header / sink (sinks: at the end)
            const int o = it.hpos;
            const unsigned rbo = w.relBits[o];
            const bool atVisitOnly = (rbo & 4u) || !(rbo & 2u);
            if ((it.bits & 1u) && ldAgent(&fx.act[o])) atomicMin(&fx.T[it.slot], w.hdrPos[o]);
            if (it.bits & 2u) {
                if (atVisitOnly) atomicMin(&fx.T[it.slot], w.hdrPos[o]);
                else stAgent(&w.items[i].id, kNever);                      // real target: `id` is free, it holds the last T sent
            }
        }
        if (!fixSync(fx, fs, anyWords, false, s.acc, nullptr)) return;
        // frozen => re-visited at once, held at its current position: T flows along the current-state entries.  Every workgroup
        // owns a contiguous stretch of the item sequence (= a slab of the mesh: the items follow the point ids) and sweeps
        // it several times between two grid barriers, so chains that stay inside a slab do not cost a barrier per link.
        // Round 3: the sweeps run out of registers and LDS.  A thread keeps its (<= kFixPer) propagating items -- owner slot,
        // target slot, last value sent -- in registers for the whole launch, and the T values of the slab's owners (a contiguous
        // slot range: the items are grouped by owner, owners descend) live in LDS between two barriers: a sweep is an LDS
        // read, a compare and an LDS or global atomicMin instead of four dependent agent-scope round trips (2.0 us per sweep
        // -> 0.3).  The slab's values are re-read from the global array at the start of every sweep (independent loads) and
        // what the slab lowered itself -- a dirty bit per slot -- is folded back at its end; values that leave the slab go out
        // with global atomics at once.  The fixed point is the same (min-propagation is order-free).
        {
            const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
            if (useLocal) {
#pragma unroll
                for (int k = 0; k < kFixPer; ++k) myLast[k] = kNever;
                for (;;) {
                    ++nVotes;
                    bool chAny = false;
                    for (int sweep = 0; sweep < maxSweeps; ++sweep) {
                        // the slab's owners as the other workgroups see them right now (they fold their findings in with global
                        // atomics sweep by sweep, as this one does below): independent loads, not a chain
                        for (int x = threadIdx.x; x < oN; x += kFixBlock) Tloc[x] = ldAgent(&fx.T[oLo + x]);
                        for (int x = threadIdx.x; x < (oN + 31) / 32; x += kFixBlock) dirty[x] = 0u;
                        __syncthreads();
                        bool ch = false;
                        for (int pass = 0; pass < 4; ++pass) {           // a few LDS-only passes: chains inside the slab
                            bool chp = false;
#pragma unroll
                            for (int k = 0; k < kFixPer; ++k) {
                                if (myO[k] < 0) continue;
                                const int t = Tloc[myO[k] - oLo];
                                if (t < myLast[k]) {
                                    const int tg = myT[k];
                                    if (tg >= oLo && tg <= oHi) { if (atomicMin(&Tloc[tg - oLo], t) > t) atomicOr(&dirty[(tg - oLo) >> 5], 1u << ((tg - oLo) & 31)); }
                                    else atomicMin(&fx.T[tg], t);
                                    myLast[k] = t;
                                    chp = true;
                                }
                            }
                            if (!fixAny(anyWords, fs, chp)) break;
                            ch = true;
                        }
                        // what the slab learned about its own owners goes back to the global array
                        for (int x = threadIdx.x; x < oN; x += kFixBlock) if ((dirty[x >> 5] >> (x & 31)) & 1u) atomicMin(&fx.T[oLo + x], Tloc[x]);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!fixAny(anyWords, fs, ch)) break;
                        chAny = true;
                    }
                    bool changed = false;
                    if (!fixSync(fx, fs, anyWords, chAny, s.acc, &changed)) { ok = false; break; }
                    if (!changed) break;
                }
            } else {
                for (;;) {
                    ++nVotes;
                    bool chAny = false;
                    for (int sweep = 0; sweep < maxSweeps; ++sweep) {
                        bool ch = false;
                        for (int i = i0 + threadIdx.x; i < i1; i += kFixBlock) {
                            const WalkItem it = w.items[i];
                            if ((it.bits & 0x80u) || (it.bits & 34u) != 34u) continue;
                            const int o = it.hpos;
                            const unsigned rbo = w.relBits[o];
                            if ((rbo & 4u) || !(rbo & 2u)) continue;
                            const int t = ldAgent(&fx.T[o]);
                            if (t < ldAgent(&w.items[i].id)) { atomicMin(&fx.T[it.slot], t); stAgent(&w.items[i].id, t); ch = true; }
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!fixAny(anyWords, fs, ch)) break;
                        chAny = true;
                    }
                    bool changed = false;
                    if (!fixSync(fx, fs, anyWords, chAny, s.acc, &changed)) { ok = false; break; }
                    if (!changed) break;
                }
            }
            tSweeps += __builtin_amdgcn_s_memrealtime() - ts0;
        }
        // who was still free at its own first visit?
        bool ch = false;
        for (int x = gtid; x < nR; x += gstride) {
            const unsigned rb = w.relBits[x];
            if ((rb & 7u) != 2u) continue;                                  // moved, own move fine, not frozen before
            const int a = ldAgent(&fx.T[x]) >= w.hdrPos[x] ? 1 : 0;
            if (a != ldAgent(&fx.act[x])) { stAgent(&fx.act[x], a); ch = true; }
        }
        if (!ok) return;
        {
            bool changed = false;
            if (!fixSync(fx, fs, anyWords, ch, s.acc, &changed)) return;
            if (!changed) break;
        }
        if (outer > 4096) { if (gtid == 0) s.acc->err = 3; break; }       // cannot happen: every round fixes a longer prefix
    }
    if (!ok) return;
    if (gtid == 0 && fx.flags[15] == 12345) {   // (debug statistics, SMGPU_WALK_STATS)
        fx.flags[8] += nOuter; fx.flags[9] += nVotes; fx.flags[10] += (int)(__builtin_amdgcn_s_memrealtime() - tStart); fx.flags[11] += (int)tSweeps; fx.flags[12] += 1;
    }
    if (fx.actPrev)
        for (int x = gtid; x < nR; x += gstride)
            if ((w.relBits[x] & 7u) == 2u) fx.actPrev[w.items[w.hdrPos[x]].id] = (uint8_t)ldAgent(&fx.act[x]);
    // results: every point that got a freeze step, and every sink an entry fired at
    for (int x = gtid; x < nR; x += gstride)
        if (!(w.relBits[x] & 4u) && ldAgent(&fx.T[x]) != kNever) s.frozen[w.items[w.hdrPos[x]].id] = 1;
    for (int i = gtid; i < nItems; i += gstride) {
        const WalkItem it = w.items[i];
        if ((it.bits & 0x80u) || (it.bits & 32u) || it.slot != nR) continue;   // sinks not frozen before the walk
        const int o = it.hpos;
        const unsigned rbo = w.relBits[o];
        const bool atVisitOnly = (rbo & 4u) || !(rbo & 2u);
        const bool fired = ((it.bits & 1u) && ldAgent(&fx.act[o])) || ((it.bits & 2u) && (atVisitOnly || ldAgent(&fx.T[o]) != kNever));
        if (fired) s.frozen[it.id] = 1;
    }
}

__global__ void __launch_bounds__(kBlock) k_walk_apply(State s, const int* ids, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) s.frozen[ids[i]] = 1;
}

}  // namespace smgpu
