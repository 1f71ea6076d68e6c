// kernels_boundary.hpp -- per-iteration part of the optional boundary point smoothing:
//
//   SM.C:2266        calculateBoundaryPointNormals OBB.C:141-233                      -> k_bnd_normals
//   SM.C:2311-2330   calculateFeatureEdgeProjections BPS.C:623-677                    -> k_bnd_feature
//                    projectBoundaryPointsToEdgesAndSurfaces BPS.C:843-945 (corners, feature edges, sharp edge
//                    freeze, projection to the target surface along +-normal with findIntersection BPS.C:682-745)
//   SM.C:2337-2351   projectPrismaticInternalPointsToSurfaces OBB.C:573-631
//   SM.C:2356        constrainMaxStepLength once more (every point: the internal ones inside the smoothing kernels)
//   then, for the boundary points, the rest of the iteration that the smoothing kernels do for the internal points:
//   restrictEdgeShortening SM.C:602-652 and -- constraints off -- restore / count / residual / move    -> k_bnd_fix
//
// The smoothing kernels leave the boundary points' proposals (centroidal + aspect-ratio blend + step clamp [+ layer
// treatment]) in State::prop and skip them; k_bnd_fix finishes them.  All of it is per-point work on a few per cent of
// the points with gathers from global memory: latency-bound kernels of a few microseconds, not HBM-bound.
//
// findLine (OpenFOAM indexedOctree<treeDataTriSurface>::findLine, third-party code absent from the reference tree): the
// intersection nearest to the start of a segment.  Every candidate triangle is tested with OpenFOAM's
// triangle::intersection(orig, dir, HALF_RAY, tol = indexedOctree::perturbTol() = 10*SMALL) (Moller-Trumbore) on the full
// segment; the smallest parameter wins, ties to the lowest triangle id -- so the result does not depend on the order
// in which the bounding volume hierarchy presents the triangles (the oracle tests all of them in id order).
#pragma once
#include "kernels.hpp"

namespace smgpu {

// per boundary point flag bits (BndView::flags)
constexpr uint8_t BF_CORNER = 1, BF_FEATURE = 2, BF_SMOOTHSURF = 4, BF_CONNECTED = 8, BF_SHARP = 16;
constexpr int BND_ERR_NORMAL = 3, BND_ERR_NOHIT = 4, BND_ERR_STRING = 5;   // Accum::err codes

struct BndView {
    int nB;                     // boundary (non-internal) points, ascending point id
    const int* pts;             // [nB] point id
    uint8_t* flags;             // [nB] BF_*; BF_SHARP is rewritten by k_bnd_normals every iteration
    const double* corner;       // [3 nB] cornerPoints (BPS.C:387-388)
    const int* inner;           // [nB] pointToInnerPointMap or -1
    const int* bfOff; const int* bfVal;   // boundary faces of the point on non-processor, non-empty patches, ascending
    const uint8_t* ptClass;     // [nPoints] BF_CORNER | BF_FEATURE of every mesh point
    int nFeat;                  // feature edge points
    const int* featPts; const int* featString; const int* featOfBnd;   // featOfBnd[nB] = index into the feature list or -1
    double* featSum; int* featCnt;        // featureEdgeProjections / nFeatureEdgeProjections of the feature points
    int nTE;                    // target edge mesh
    const double* tePts; const int* teEdges; const int* teString;
    const int* strOff; const int* strEdges;   // CSR string -> its edges (ascending)
    int nNodes;                 // 8-wide bounding volume hierarchy over the target triangles (boundary.hpp)
    const float* wideBox;       // 48 floats per node (see findLine)
    const int* wideRef;         // 16 ints per node: ref[8], cnt[8]
    const double* triVerts;     // 10 doubles per triangle in leaf order: nine coordinates + the original id (as bits)
    double distanceTolerance, internalBlend;
};

// OpenFOAM face area vector (primitiveMesh::makeFaceCentresAndAreas, as k_face_geom)
__device__ __forceinline__ V3 faceAreaOf(const MeshView& m, const double* __restrict__ P, int f) {
    const int b = m.faceOff[f];
    const int n = m.faceOff[f + 1] - b;
    if (n == 3) {
        const V3 p0 = ldv(P, m.facePts[b]), p1 = ldv(P, m.facePts[b + 1]), p2 = ldv(P, m.facePts[b + 2]);
        return 0.5 * cross(p1 - p0, p2 - p0);
    }
    V3 fCentre = ldv(P, m.facePts[b]);
    for (int i = 1; i < n; ++i) fCentre = fCentre + ldv(P, m.facePts[b + i]);
    fCentre = fCentre / double(n);
    V3 sumN = v3(0, 0, 0);
    double sumA = 0.0;
    V3 thisPoint = ldv(P, m.facePts[b]);
    const V3 first = thisPoint;
    for (int i = 0; i < n; ++i) {
        const V3 nextPoint = (i == n - 1) ? first : ldv(P, m.facePts[b + i + 1]);
        const V3 nn = cross(nextPoint - thisPoint, fCentre - thisPoint);
        sumN = sumN + nn;
        sumA += mag(nn);
        thisPoint = nextPoint;
    }
    if (sumA < SMGPU_ROOTVSMALL) return v3(0, 0, 0);
    return 0.5 * sumN;
}

// calculateBoundaryPointNormals OBB.C:141-233 for the boundary points.  The reference never resets pointNormals: the
// new normal is the normalised sum of the previous (unit) normal and the inverted unit normals of the point's
// boundary faces.  (Internal points keep copies made at set-up; layerTreat re-normalises those.)
__device__ __forceinline__ void bndNormalsOf(const MeshView& m, const State& s, const BndView& b, int i) {
    if (s.acc->stop) return;
    if (i >= b.nB) return;
    const int p = b.pts[i];
    V3 n = ldv(s.layerNormal, p);
    const int f0 = b.bfOff[i], f1 = b.bfOff[i + 1];
    for (int k = f0; k < f1; ++k) {
        const V3 cSf = faceAreaOf(m, s.ptsCur, b.bfVal[k]);
        n = n - cSf / mag(cSf);
    }
    // multi-rank: a shared point's local sum goes to the other sharers first (plusEq, OBB.C:184-198); k_bnd_normals_shared
    // classifies and normalises it after the exchange
    if (s.sharedSlot && s.sharedSlot[p] >= 0) { stv(s.layerNormal, p, n); return; }
    uint8_t fl = b.flags[i];
    if (f1 > f0) {
        if (mag(n) < 0.1) { n = v3(0, 0, 0); fl |= BF_SHARP; }
        else fl &= (uint8_t)~BF_SHARP;
        b.flags[i] = fl;
    }
    if (n != v3(0, 0, 0)) n = n / mag(n);
    stv(s.layerNormal, p, n);
}
__global__ void __launch_bounds__(kBlock) k_bnd_normals(MeshView m, State s, BndView b) { bndNormalsOf(m, s, b, blockIdx.x * kBlock + threadIdx.x); }

// OBB.C:201-230 for the shared boundary points, on the sums over the sharers (combL: normal [0:3], face count [6])
__device__ __forceinline__ void bndNormalsSharedOf(const State& s, const BndView& b, int nShared, const int* sharedLocal, int i) {
    if (s.acc->stop) return;
    if (i >= nShared) return;
    const int bi = s.bndOfShared[i];
    if (bi < 0) return;
    const double* r = s.combL + (size_t)i * s.lStride;
    V3 n = v3(r[0], r[1], r[2]);
    uint8_t fl = b.flags[bi];
    if (r[6] >= 1.0) {
        if (mag(n) < 0.1) { n = v3(0, 0, 0); fl |= BF_SHARP; }
        else fl &= (uint8_t)~BF_SHARP;
        b.flags[bi] = fl;
    }
    if (n != v3(0, 0, 0)) n = n / mag(n);
    stv(s.layerNormal, sharedLocal[i], n);
}
__global__ void __launch_bounds__(kBlock) k_bnd_normals_shared(State s, BndView b, int nShared, const int* sharedLocal) {
    bndNormalsSharedOf(s, b, nShared, sharedLocal, blockIdx.x * kBlock + threadIdx.x);
}
// exchange A's combine (k_halo_combineA2) and, in the workgroups after its nA, exchange L's (k_halo_combineL) followed for the
// same point by OBB.C:201-230 (k_bnd_normals_shared) -- three launch-latency-bound launches in one
__global__ void __launch_bounds__(kBlock, 2) k_halo_combineAL(int nShared, const int* __restrict__ peer, const double* __restrict__ ownA,
                                                           const double* __restrict__ recvA, double* __restrict__ combA, int nBlocksTwo, int nMulti,
                                                           const int* multiIdx, const int* multiSlots, int nA, State s, BndView b, int bndOn,
                                                           const int* combOff, const int* combSlots, const double* ownL, const double* recvL,
                                                           double* combL, const int* sharedLocal, PushWait pw) {
    pushWait(pw);
    const int bx = (int)blockIdx.x;
    if (bx < nA) { haloCombineA2Of(bx, nShared, peer, ownA, recvA, combA, nBlocksTwo, nMulti, multiIdx, multiSlots, s.ownFold); return; }
    const int i = (bx - nA) * kBlock + (int)threadIdx.x;
    haloCombineLOf(i, nShared, combOff, combSlots, ownL, recvL, combL, s.lStride, s.ownFold);
    if (bndOn) bndNormalsSharedOf(s, b, nShared, sharedLocal, i);     // reads the record this thread has just written
}

// projectPointToEdge BPS.C:89-145 (the edge point index it also reports is not consumed per iteration)
__device__ __forceinline__ V3 projectToEdge(const BndView& b, const V3& pt, int e) {
    const V3 a = ldv(b.tePts, b.teEdges[2 * e]), c = ldv(b.tePts, b.teEdges[2 * e + 1]);
    const double edgeLength = mag(c - a);
    const V3 c2pt = pt - a, edgeVec = c - a;
    const double sN = dot(c2pt, edgeVec) / (edgeLength * edgeLength);
    if (sN <= 1e-6) return a;                 // ABS_TOL, COM.H:21
    if (sN >= (1.0 - 1e-6)) return c;
    return a + sN * edgeVec;
}

// calculateFeatureEdgeProjections BPS.C:623-677: one wave per feature edge point; for each eligible neighbour the
// lanes share the scan over the target edges of the point's string (findClosestEdgeInfo BPS.C:206-264: strict "<",
// so the lowest edge id among equal distances) and lane 0 accumulates in pointPoints order.
__device__ __forceinline__ void bndFeatureOf(const MeshView& m, const State& s, const BndView& b, int j, int lane) {
    if (s.acc->stop) return;
    if (j >= b.nFeat) return;
    const int p = b.featPts[j], str = b.featString[j];
    V3 sum = v3(0, 0, 0);
    int cnt = 0;
    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
        const int q = m.ppPt[k];
        if (m.pflags[q] & PF_INTERNAL) continue;          // findNeighborSurfacePoints BPS.C:592-615
        if (b.ptClass[q] & (BF_FEATURE | BF_CORNER)) continue;
        const V3 pt = ldv(s.ptsCur, q);
        double best = SMGPU_GREAT;
        int bestE = 0x7fffffff;
        V3 bestP = v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT);
        // the edges of the point's string, in ascending id order (all edges when the point has no string)
        const int e0 = (str >= 0) ? b.strOff[str] : 0, e1 = (str >= 0) ? b.strOff[str + 1] : b.nTE;
        for (int ke = e0 + lane; ke < e1; ke += 64) {
            const int e = (str >= 0) ? b.strEdges[ke] : ke;
            const V3 pr = projectToEdge(b, pt, e);
            const double d = mag(pr - pt);
            if (d < best) { best = d; bestE = e; bestP = pr; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(best, o, 64);
            const int oe = __shfl_xor(bestE, o, 64);
            const V3 op = v3(__shfl_xor(bestP.x, o, 64), __shfl_xor(bestP.y, o, 64), __shfl_xor(bestP.z, o, 64));
            if (od < best || (od == best && oe < bestE)) { best = od; bestE = oe; bestP = op; }
        }
        if (bestE == 0x7fffffff) { if (lane == 0) s.acc->err = BND_ERR_STRING; continue; }   // BPS.C:258-261
        sum = sum + bestP;
        ++cnt;
    }
    if (lane == 0) { stv(b.featSum, j, sum); b.featCnt[j] = cnt; }
}
__global__ void __launch_bounds__(64) k_bnd_feature(MeshView m, State s, BndView b) { bndFeatureOf(m, s, b, (int)blockIdx.x, (int)threadIdx.x); }

// The geometry launch with the two kernels above riding in its FIRST workgroups (they only read the current coordinates, as
// the geometry does, and are latency-bound chains over a few ten thousand boundary points): a side stream for them cost
// ~38 us of fork / join idle time per iteration around 23 + 14 us of kernels (rocprofv3 trace of the multi-rank probe with
// boundary point smoothing); here they cost no launch at all.  nBndBlocks (a multiple of 8, so that the XCD mapping of the
// geometry workgroups is unchanged) = normals blocks, then feature blocks (T / 64 feature points each).
template <int T, bool ORG>
__global__ void __launch_bounds__(T, (SMGPU_GEOM_WAVES * T) / 256) k_geom_tile_bnd(MeshView m, State s, GeomTileView g, int wantAvg, int writeFaces,
                                                  const int* tileList, int nLaunch, int xcdMap, int deferN, int deferIter, double* deferLocal,
                                                  double* deferHist, BndView b, int nBndBlocks, int nNormalBlocks) {
    const int bx = (int)blockIdx.x;
    if (bx < nBndBlocks) {
        if (bx < nNormalBlocks) bndNormalsOf(m, s, b, bx * T + (int)threadIdx.x);
        else bndFeatureOf(m, s, b, (bx - nNormalBlocks) * (T / 64) + ((int)threadIdx.x >> 6), (int)threadIdx.x & 63);
        return;
    }
    geomTileBody<T, ORG>(m, s, g, wantAvg, writeFaces, tileList, nLaunch, xcdMap, deferN, deferIter, deferLocal, deferHist, bx - nBndBlocks);
}

// OpenFOAM triangle::intersection(orig, dir, intersection::HALF_RAY, tol)
__device__ __forceinline__ bool triangleIntersection(const double* __restrict__ tv, const V3& orig, const V3& dir, double tol, double& t, V3& pt) {
    const V3 a = v3(tv[0], tv[1], tv[2]), bb = v3(tv[3], tv[4], tv[5]), c = v3(tv[6], tv[7], tv[8]);
    const V3 edge1 = bb - a, edge2 = c - a;
    const V3 pVec = cross(dir, edge2);
    const double det = dot(edge1, pVec);
    if (det > -SMGPU_ROOTVSMALL && det < SMGPU_ROOTVSMALL) return false;
    const double inv_det = 1.0 / det;
    const V3 tVec = orig - a;
    const double u = dot(tVec, pVec) * inv_det;
    if (u < -tol || u > 1.0 + tol) return false;
    const V3 qVec = cross(tVec, edge1);
    const double v = dot(dir, qVec) * inv_det;
    if (v < -tol || u + v > 1.0 + tol) return false;
    t = dot(edge2, qVec) * inv_det;
    if (t < -tol) return false;
    pt = a + u * edge1 + v * edge2;
    return true;
}

// One 8-wide node: the child boxes as floats rounded outwards (conservative), structure-of-arrays so that a node is
// twelve 16-byte loads: lo.x[8], lo.y[8], lo.z[8], hi.x[8], hi.y[8], hi.z[8]  (192 bytes, 64-byte aligned rows).
// The traversal stack lives in LDS, one column per thread ([entry][thread]: conflict-free): in private memory every
// push / pop would be a scratch round trip in the dependent chain of the descent.  7 pending siblings per level of the
// 8-wide tree; the host checks 7 * depth + 1 <= kBvhStack (depth 9: far beyond any surface that fits the device).
constexpr int kBvhStack = 64;

__device__ __forceinline__ bool findLine(const BndView& b, int* __restrict__ stack, const V3& start, const V3& end, V3& hitPoint) {
    const V3 dir = end - start;
    const double tol = 10.0 * 1.0e-15;   // indexedOctree::perturbTol() = 10*SMALL
    // Slab test per axis, t = box * (1/d) - o * (1/d) as one fused multiply-add (this is culling, not reference
    // arithmetic), near / far plane picked by the sign of the direction.  The parameter range carries the slack
    // (rounding here is ~1e-16 |o/d|, the boxes are inflated besides); an axis the segment does not move along only
    // checks the origin.  The test may accept too much, never too little.
    const double o3[3] = {start.x, start.y, start.z}, d3[3] = {dir.x, dir.y, dir.z};
    double inv[3], oinv[3];
    bool flat[3], neg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        flat[a] = d3[a] == 0.0;
        neg[a] = d3[a] < 0.0;
        inv[a] = flat[a] ? 0.0 : 1.0 / d3[a];
        oinv[a] = -(o3[a] * inv[a]);
    }
    double best = 0.0;
    int bestId = 0x7fffffff;
    // One kind of work per loop trip: a popped node tests its eight child boxes and pushes what the segment touches
    // (leaves too, as -(16 * first + count) - 1); a popped leaf loads its (at most four) triangles in one batch and
    // tests them.  Lanes of a wave that are at different places of their descents still meet at these two program
    // points, instead of serialising leaf visits inside the child loop.
    // boxes entered beyond the best hit so far cannot hold a nearer (or equal, lower-id) one: the limit follows the hits
    double tLimit = 1.0 + 1e-6;
    int sp = 0;
    if (b.nNodes > 0) stack[kBlock * sp++] = 0;
    while (sp > 0) {
        const int item = stack[kBlock * --sp];
        if (item >= 0) {
            const float4* __restrict__ nb = reinterpret_cast<const float4*>(b.wideBox) + 12 * (size_t)item;
            float bx[48];
#pragma unroll
            for (int q = 0; q < 12; ++q) { const float4 v = nb[q]; bx[4 * q] = v.x; bx[4 * q + 1] = v.y; bx[4 * q + 2] = v.z; bx[4 * q + 3] = v.w; }
            const int4* __restrict__ rc = reinterpret_cast<const int4*>(b.wideRef) + 4 * (size_t)item;   // ref[8], cnt[8]
            const int4 r0 = rc[0], r1 = rc[1], c0 = rc[2], c1 = rc[3];
            const int ref[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const int cnt[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = cnt[c];
                double t0 = -1e-6, t1 = tLimit;
                bool touch = n >= 0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double lo = (double)bx[8 * a + c], hi = (double)bx[24 + 8 * a + c];
                    if (flat[a]) { touch = touch && !(o3[a] < lo || o3[a] > hi); }
                    else {
                        const double ta = fma(neg[a] ? hi : lo, inv[a], oinv[a]), tb = fma(neg[a] ? lo : hi, inv[a], oinv[a]);
                        t0 = (ta > t0) ? ta : t0;
                        t1 = (tb < t1) ? tb : t1;
                    }
                }
                if (touch && !(t0 > t1) && sp < kBvhStack) stack[kBlock * sp++] = (n == 0) ? ref[c] : -(16 * ref[c] + n) - 1;
            }
        } else {
            const int code = -(item + 1), first = code >> 4, n = code & 15;
            // a triangle record is 80 bytes = five 16-byte loads: nine coordinates and the triangle id (the memory
            // pipeline's cost here is the number of load instructions per lane, not the bytes)
            double tv[4][9];
            int ids[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < n) {
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(b.triVerts) + 5 * (size_t)(first + k);
                    const double2 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4];
                    tv[k][0] = q0.x; tv[k][1] = q0.y; tv[k][2] = q1.x; tv[k][3] = q1.y; tv[k][4] = q2.x; tv[k][5] = q2.y;
                    tv[k][6] = q3.x; tv[k][7] = q3.y; tv[k][8] = q4.x;
                    ids[k] = (int)__double_as_longlong(q4.y);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double t;
                V3 pt;
                if (k >= n || !triangleIntersection(tv[k], start, dir, tol, t, pt)) continue;
                if (!(t <= 1.0)) continue;   // treeDataTriSurface::findIntersectOp: inter.distance() <= 1
                if (bestId == 0x7fffffff || t < best || (t == best && ids[k] < bestId)) { best = t; bestId = ids[k]; hitPoint = pt; }
            }
            if (bestId != 0x7fffffff) tLimit = fmin(tLimit, best + 1e-6 * (1.0 + fabs(best)));
        }
    }
    return bestId != 0x7fffffff;
}

// findIntersection BPS.C:682-745.  Called by BOTH lanes of a pair (role 0 / 1, lanes 2k and 2k + 1 work on the same
// point): the two half-ray queries run side by side, one per lane, and the results are exchanged by shuffles.
__device__ __forceinline__ V3 findIntersectionPair(const BndView& b, int* __restrict__ stack, const V3& origPoint, const V3& pointNormal, double searchDistance, int role) {
    const V3 undef = v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT);
    const V3 endPoint1 = origPoint + pointNormal * searchDistance;
    const V3 endPoint2 = origPoint - pointNormal * searchDistance;
    V3 mine = undef, h;
    if (findLine(b, stack, origPoint, role ? endPoint2 : endPoint1, h)) mine = h;
    const V3 other = v3(__shfl_xor(mine.x, 1, 64), __shfl_xor(mine.y, 1, 64), __shfl_xor(mine.z, 1, 64));
    const V3 hitPoint1 = sel3(role, other, mine), hitPoint2 = sel3(role, mine, other);
    const double distance1 = mag(origPoint - hitPoint1);
    const double distance2 = mag(origPoint - hitPoint2);
    if (distance1 < distance2) return hitPoint1;
    else if (distance2 < distance1) return hitPoint2;
    if (findLine(b, stack, endPoint1, endPoint2, h)) return h;
    return undef;
}

// The boundary points' part of the iteration from the projection on (see the file header).
// Two lanes per boundary point: they differ only in which half-ray they trace (findIntersectionPair); lane "role 0"
// writes the results.
template <bool FINAL>
__global__ void __launch_bounds__(kBlock) k_bnd_fix(MeshView m, State s, Prm prm, BndView b, int partialBase) {
    if (s.acc->stop) return;
    __shared__ int bvhStack[kBvhStack * kBlock];
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int i = t >> 1, role = t & 1;
    double dist = 0.0;
    int fcount = 0;
    if (i < b.nB) {
        const int p = b.pts[i];
        const uint8_t fl = b.flags[i];
        const V3 cur = ldv(s.ptsCur, p);
        V3 np = ldv(s.prop, p);
        const V3 undef = v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT);
        bool frozen = false;
        // multi-rank: a shared point takes the values combined over its sharers (BPS.C:659-674, OBB.C:490-496)
        const int slot = s.sharedSlot ? s.sharedSlot[p] : -1;
        const double* cl = (slot >= 0) ? s.combL + (size_t)slot * s.lStride : nullptr;
        // projectBoundaryPointsToEdgesAndSurfaces BPS.C:876-940
        if (fl & BF_CORNER) np = ldv(b.corner, i);
        else if (fl & BF_FEATURE) {
            const int j = b.featOfBnd[i];
            if (cl) np = v3(cl[10], cl[11], cl[12]) / cl[13];
            else np = ldv(b.featSum, j) / double(b.featCnt[j]);
        } else if (fl & BF_SHARP) frozen = true;
        else if (fl & BF_SMOOTHSURF) {
            const V3 pointNormal = ldv(s.layerNormal, p);
            if (pointNormal == v3(0, 0, 0)) s.acc->err = BND_ERR_NORMAL;   // BPS.C:691-696
            else {
                double searchDistance = b.distanceTolerance;
                V3 surfPoint = undef;
                for (int it = 0; it < 4; ++it) {
                    searchDistance *= (1.0 / 1e-4);   // 1.0 / REL_TOL
                    surfPoint = findIntersectionPair(b, bvhStack + threadIdx.x, np, pointNormal, searchDistance, role);
                    if (surfPoint != undef) { np = surfPoint; break; }
                }
                if (surfPoint == undef) s.acc->err = BND_ERR_NOHIT;        // BPS.C:932-938
            }
        }
        if (role == 0) {   // the second lane of the pair was only needed for its half-ray
        // projectPrismaticInternalPointsToSurfaces OBB.C:573-631
        if ((fl & BF_SMOOTHSURF) && (fl & BF_CONNECTED) && b.inner[i] >= 0 && !(fl & (BF_FEATURE | BF_CORNER | BF_SHARP))) {
            const V3 pointNormal = ldv(s.layerNormal, p);
            if (pointNormal == v3(0, 0, 0)) s.acc->err = BND_ERR_NORMAL;
            const V3 innerNeighCoord = cl ? v3(cl[7], cl[8], cl[9]) : ldv(s.ptsCur, b.inner[i]);   // updateNeighCoords OBB.C:464-500
            const V3 cCoords = np;
            const V3 neighVec = cCoords - innerNeighCoord;
            const double dotProd = dot(neighVec, pointNormal);
            const V3 pVec = neighVec - dotProd * pointNormal;
            const V3 newCoords = cCoords - pVec;
            np = b.internalBlend * newCoords + (1.0 - b.internalBlend) * np;
        }
        {   // SM.C:2356 constrainMaxStepLength
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        if (!frozen) {   // restrictEdgeShortening SM.C:611-648 (skips the points frozen above)
            double shortestCur = SMGPU_GREAT, shortestNew = SMGPU_GREAT;
            for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
                const V3 nb = ldv(s.ptsCur, m.ppPt[k]);
                const double tc = mag(cur - nb);
                if (tc < shortestCur) shortestCur = tc;
                const double tn = mag(np - nb);
                if (tn < shortestNew) shortestNew = tn;
            }
            const double shortest = (shortestNew < shortestCur) ? shortestNew : shortestCur;
            if (prm.totalMinFreeze && (shortest < prm.minEdge)) frozen = true;
            else if ((shortestNew < prm.minEdge) && (shortestNew < shortestCur)) frozen = true;
        }
        if (FINAL && slot < 0) {
            if (frozen || !(m.pflags[p] & PF_SMOOTHSURF)) { np = cur; fcount = 1; }   // SM.C:2384-2392
            dist = mag(np - cur) / prm.maxStep;
            stv(s.ptsNext, p, np);
        } else {
            stv(s.prop, p, np);
            s.frozen[p] = frozen ? 1 : 0;
        }
        }
    }
    if (FINAL) blockPublish<kBlock>(s, dist, fcount, partialBase + blockIdx.x);
}

// parity access to the segment query: one thread per segment
__global__ void __launch_bounds__(kBlock) k_bnd_find_line(BndView b, int n, const double* seg, double* out, int* hit) {
    __shared__ int bvhStack[kBvhStack * kBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    V3 h = v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT);
    hit[i] = findLine(b, bvhStack + threadIdx.x, ldv(seg, 2 * i), ldv(seg, 2 * i + 1), h) ? 1 : 0;
    stv(out, i, h);
}

}  // namespace smgpu
