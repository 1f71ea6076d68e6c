// kernels_walk.hpp -- compacted form of the face-angle freeze walk (SM.C:1347-1434) for meshes where
// many points lie outside the good angle range (e.g. refinement interfaces: coplanar face pairs give
// face angles of 180 degrees in every iteration).
//
// The walk itself is inherently sequential (LIFO stack, reads and writes isFrozenPoint as it goes), but
// everything expensive in it is a pure function of (current coordinates, proposals).  So:
//   1. k_walk_count / k_walk_scan / k_walk_fill : ordered compaction of the active points (ascending
//      point id) and of their pointPoints rows into dense tables;
//   2. k_walk_pred : one thread per table entry evaluates the geometric predicates in parallel
//      (self-deterioration, and for every neighbour "its move hurts me" with me at my proposal / at my
//      current position) -- the same bits as k_fa_pred;
//   3. the host replays the reference's stack order over the bit tables (a few ns per entry; the same
//      replay on one GPU lane costs microseconds per point), and
//   4. k_walk_apply marks the points the replay froze.
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

__global__ void __launch_bounds__(kBlock) k_walk_count(MeshView m, State s, WalkView w) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    int a = 0, e = 0;
    if (p < m.nPoints && s.faActive[p]) { a = 1; e = m.ppOff[p + 1] - m.ppOff[p]; }
    __shared__ int sa[kBlock / 64], se[kBlock / 64];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); e += __shfl_down(e, o, 64); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; se[threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ta = 0, te = 0;
        for (int i = 0; i < kBlock / 64; ++i) { ta += sa[i]; te += se[i]; }
        w.blkA[blockIdx.x] = ta;
        w.blkE[blockIdx.x] = te;
    }
}

// exclusive scan of the per-block counts (one workgroup; nBlk is at most a few 10^4)
__global__ void __launch_bounds__(kBlock) k_walk_scan(State s, WalkView w, int nBlk) {
    if (s.acc->stop) return;
    __shared__ int baseA, baseE;
    __shared__ int wa[kBlock / 64], we[kBlock / 64];
    if (threadIdx.x == 0) { baseA = 0; baseE = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < nBlk; b0 += kBlock) {
        const int i = b0 + threadIdx.x;
        const int a = (i < nBlk) ? w.blkA[i] : 0, e = (i < nBlk) ? w.blkE[i] : 0;
        int ia = a, ie = e;   // inclusive scan inside the wave
        const int lane = threadIdx.x & 63;
        for (int o = 1; o < 64; o <<= 1) {
            const int ta = __shfl_up(ia, o, 64), te = __shfl_up(ie, o, 64);
            if (lane >= o) { ia += ta; ie += te; }
        }
        if (lane == 63) { wa[threadIdx.x >> 6] = ia; we[threadIdx.x >> 6] = ie; }
        __syncthreads();
        int offA = baseA, offE = baseE;
        for (int k = 0; k < (threadIdx.x >> 6); ++k) { offA += wa[k]; offE += we[k]; }
        if (i < nBlk) { w.blkA[i] = offA + ia - a; w.blkE[i] = offE + ie - e; }
        __syncthreads();
        if (threadIdx.x == kBlock - 1) { baseA = offA + ia; baseE = offE + ie; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { w.header[0] = baseA; w.header[1] = baseE; }
}

__global__ void __launch_bounds__(kBlock) k_walk_fill(MeshView m, State s, WalkView w) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    const bool act = p < m.nPoints && s.faActive[p];
    const int a = act ? 1 : 0, e = act ? m.ppOff[p + 1] - m.ppOff[p] : 0;
    int ia = a, ie = e;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(ia, o, 64), te = __shfl_up(ie, o, 64);
        if (lane >= o) { ia += ta; ie += te; }
    }
    __shared__ int wa[kBlock / 64], we[kBlock / 64];
    if (lane == 63) { wa[threadIdx.x >> 6] = ia; we[threadIdx.x >> 6] = ie; }
    __syncthreads();
    int offA = w.blkA[blockIdx.x], offE = w.blkE[blockIdx.x];
    for (int k = 0; k < (threadIdx.x >> 6); ++k) { offA += wa[k]; offE += we[k]; }
    if (p < m.nPoints) w.activeSlot[p] = act ? offA + ia - 1 : -1;
    if (act) {
        const int slot = offA + ia - 1, eo = offE + ie - e;
        w.actIds[slot] = p;
        w.actEntOff[slot] = eo;
        const int nb = m.ppOff[p];
        for (int j = 0; j < e; ++j) { w.entOwner[eo + j] = slot; w.entNbr[eo + j] = m.ppPt[nb + j]; }
    }
}

// predicates, one thread per (active point, neighbour) entry plus one per active point (self test)
__global__ void __launch_bounds__(kBlock) k_walk_pred(MeshView m, State s, Prm prm, WalkView w, int nA, int nE) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= nA + nE) return;
    if (t == 0) w.actEntOff[nA] = nE;
    const int slot = (t < nA) ? t : w.entOwner[t - nA];
    const int p = w.actIds[slot];
    const V3 cur = ldv(s.ptsCur, p);
    const V3 np = ldv(s.prop, p);
    const double curMin = s.ptMin[p], curMax = s.ptMax[p];
    const bool moved = (np != cur);
    double mn, mx;
    if (t < nA) {
        uint8_t sb = (moved ? 2 : 0) | (s.frozen[p] ? 4 : 0);
        if (moved) {   // SM.C:1385-1394
            pointFaceAngles(m, s, p, np, -1, np, mn, mx);
            if (((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax))) sb |= 1;
        }
        w.actBits[slot] = sb;
    } else {
        const int e = t - nA;
        const int q = w.entNbr[e];
        const V3 nq = ldv(s.prop, q);
        uint8_t nb = s.frozen[q] ? 8 : 0;
        if (nq != ldv(s.ptsCur, q)) {   // SM.C:1414: the neighbour is moving
            nb |= 4;
            pointFaceAngles(m, s, p, cur, q, nq, mn, mx);   // this point held at its current position
            const bool badF = ((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax));
            if (badF) nb |= 2;
            if (moved) {
                pointFaceAngles(m, s, p, np, q, nq, mn, mx);   // this point at its proposal, SM.C:1419
                if (((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax))) nb |= 1;
            } else if (badF) nb |= 1;
        }
        w.entBits[e] = nb;
        w.entSlot[e] = w.activeSlot[q];
    }
}

// ---- second compaction (over active slots) ------------------------------------------------------------------
__device__ __forceinline__ bool entryActs(uint8_t nb) { return (nb & 4) && (nb & 3); }   // moving neighbour, hurt in some state

__global__ void __launch_bounds__(kBlock) k_rel_count(WalkView w, int nA) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    int r = 0, b = 0;
    if (a < nA) {
        for (int k = w.actEntOff[a]; k < w.actEntOff[a + 1]; ++k) b += entryActs(w.entBits[k]) ? 1 : 0;
        const uint8_t sb = w.actBits[a];
        r = (b > 0 || ((sb & 1) && (sb & 2))) ? 1 : 0;
        if (!r) b = 0;
    }
    __shared__ int sa[kBlock / 64], se[kBlock / 64];
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_down(r, o, 64); b += __shfl_down(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = r; se[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ta = 0, te = 0;
        for (int i = 0; i < kBlock / 64; ++i) { ta += sa[i]; te += se[i]; }
        w.blkA[blockIdx.x] = ta;
        w.blkE[blockIdx.x] = te;
    }
}

__global__ void __launch_bounds__(kBlock) k_rel_fill(WalkView w, int nA, int nR, int nB) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    int r = 0, b = 0;
    if (a < nA) {
        for (int k = w.actEntOff[a]; k < w.actEntOff[a + 1]; ++k) b += entryActs(w.entBits[k]) ? 1 : 0;
        const uint8_t sb = w.actBits[a];
        r = (b > 0 || ((sb & 1) && (sb & 2))) ? 1 : 0;
        if (!r) b = 0;
    }
    int ir = r, ib = b;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const int tr = __shfl_up(ir, o, 64), tb = __shfl_up(ib, o, 64);
        if (lane >= o) { ir += tr; ib += tb; }
    }
    __shared__ int wr[kBlock / 64], wb[kBlock / 64];
    if (lane == 63) { wr[threadIdx.x >> 6] = ir; wb[threadIdx.x >> 6] = ib; }
    __syncthreads();
    int offR = w.blkA[blockIdx.x], offB = w.blkE[blockIdx.x];
    for (int k = 0; k < (threadIdx.x >> 6); ++k) { offR += wr[k]; offB += wb[k]; }
    if (a < nA) w.relSlot[a] = r ? offR + ir - 1 : -1;
    if (r) {
        const int slot = offR + ir - 1;
        const int eoEnd = offB + ib;                       // entries of slots 0..slot
        // visiting order: slot nR-1 first; every earlier-visited point contributes its header and its entries
        int pos = (nR - 1 - slot) + (nB - eoEnd);
        w.hdrPos[slot] = pos;
        const int hdr = pos++;
        uint8_t anyOld = 0;
        for (int k = w.actEntOff[a]; k < w.actEntOff[a + 1]; ++k) {
            const uint8_t nb = w.entBits[k];
            if (!entryActs(nb)) continue;
            w.items[pos] = WalkItem{w.entSlot[k], w.entNbr[k], nb, 0};   // active slot for now, see k_rel_link
            anyOld |= nb & 2;
            ++pos;
        }
        // bit 3: a re-visit of this point after a neighbour froze it (held at its current position, SM.C:1431)
        // can freeze somebody; without it the replay does not need to come back
        const uint8_t rb = w.actBits[a] | (anyOld ? 8 : 0);
        w.relBits[slot] = rb;
        const bool moved = rb & 2, selfBad = rb & 1;
        w.items[hdr] = WalkItem{slot, w.actIds[a], ((moved && selfBad) ? 3u : 0u) | 32u | ((moved && !selfBad) ? 64u : 0u) | 128u, hdr};
    }
}

__global__ void __launch_bounds__(kBlock) k_rel_link(WalkView w, int nItems, int nR) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nItems) return;
    WalkItem it = w.items[i];
    if (it.bits & 0x80u) return;
    // a neighbour that is not active, or active but unable to act, is a sink: the replay only honours its
    // "frozen before the walk" bit (two constant pseudo slots behind the real ones)
    const unsigned nb = it.bits;
    const int rs = (it.slot >= 0) ? w.relSlot[it.slot] : -1;
    unsigned bits = nb & 3u;
    if (rs >= 0) {
        it.slot = rs;
        it.hpos = w.hdrPos[rs];
        bits |= 32u;
        if (w.relBits[rs] & 8) bits |= 16u;
    } else {
        it.slot = nR + ((nb >> 3) & 1u);
    }
    it.bits = bits;
    w.items[i] = it;
}

__global__ void __launch_bounds__(kBlock) k_walk_apply(State s, const int* ids, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) s.frozen[ids[i]] = 1;
}

}  // namespace smgpu
