// smooth_oracle_boundary.cpp -- CPU restatement of the reference's optional boundary point smoothing:
// projection of boundary points to feature edges and boundary surfaces (BPS.C = src/boundaryPointSmoothing.C),
// the prismatic projection of OBB.C:573-631 (OBB.C = src/orthogonalBoundaryBlending.C) and their set-up in
// SM.C:2080-2253.  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see smooth_oracle.hpp).  Serial runs only.
//
// Third-party code this path calls that is NOT under the reference tree, restated from OpenFOAM (v2412):
//  * edgeMesh::pointEdges (invertManyToMany: per point the edge ids in ascending order);
//  * indexedOctree<treeDataTriSurface>::findLine: the nearest intersection to the start point along a segment.
//    The octree's traversal order is not reproduced; every triangle is tested with OpenFOAM's
//    triangle::intersection(orig, dir, HALF_RAY, tol = indexedOctree::perturbTol() = 10*SMALL) (Moller-Trumbore) on
//    the full segment and the hit with the smallest parameter wins, ties to the lowest triangle id.
#include <algorithm>
#include <cmath>

#include "oracle_vec.hpp"
#include "smooth_oracle.hpp"

namespace orc {

static constexpr double REL_TOL = 1e-4;   // COM.H:20
static constexpr double ABS_TOL = 1e-6;   // COM.H:21
static constexpr int UNDEF_LABEL = -1;    // COM.H:14

void EdgeMesh::buildPointEdges() {
    pointEdges.assign(points.size(), {});
    for (size_t e = 0; e < edges.size(); ++e) {
        pointEdges[(size_t)edges[e][0]].push_back((int)e);
        if (edges[e][1] != edges[e][0]) pointEdges[(size_t)edges[e][1]].push_back((int)e);
    }
}

// BPS.C:20-79; returns the FatalError text, empty when sane
static std::string checkEdgeMeshSanity(const EdgeMesh& em, double meshMinEdgeLength, double meshPerimeter) {
    double minEdgeLength = VGREAT;
    double bbMinX = VGREAT, bbMaxX = -VGREAT, bbMinY = VGREAT, bbMaxY = -VGREAT, bbMinZ = VGREAT, bbMaxZ = -VGREAT;
    for (const auto& e : em.edges) {
        const Vec3 startPoint = em.points[(size_t)e[0]];
        const Vec3 endPoint = em.points[(size_t)e[1]];
        const double edgeLength = mag(endPoint - startPoint);
        if (edgeLength < minEdgeLength) minEdgeLength = edgeLength;
        for (const Vec3& q : {startPoint, endPoint}) {
            if (q.x < bbMinX) bbMinX = q.x;
            if (q.y < bbMinY) bbMinY = q.y;
            if (q.z < bbMinZ) bbMinZ = q.z;
            if (q.x > bbMaxX) bbMaxX = q.x;
            if (q.y > bbMaxY) bbMaxY = q.y;
            if (q.z > bbMaxZ) bbMaxZ = q.z;
        }
    }
    if (minEdgeLength < REL_TOL * meshMinEdgeLength) return "Minimum edge length in edge mesh is too small in comparison to minimum edge length in polyMesh";
    const double emPerimeter = bbMaxX - bbMinX + bbMaxY - bbMinY + bbMaxZ + bbMinZ;   // "+ bbMinZ" as in BPS.C:69
    const double PERIMETER_TOLERANCE = 0.5;
    if (std::abs((emPerimeter / meshPerimeter) - 1.0) > PERIMETER_TOLERANCE) return "Perimeter of edge mesh is too different in comparison to perimeter of polyMesh";
    return "";
}

// BPS.C:89-145
static void projectPointToEdge(const Vec3& pt, const EdgeMesh& em, int edgeI, double distanceTolerance, Vec3& projPoint, int& edgePointI) {
    edgePointI = UNDEF_LABEL;
    const int startPointI = em.edges[(size_t)edgeI][0];
    const int endPointI = em.edges[(size_t)edgeI][1];
    const Vec3 startPoint = em.points[(size_t)startPointI];
    const Vec3 endPoint = em.points[(size_t)endPointI];
    const double edgeLength = mag(endPoint - startPoint);
    const Vec3 c2pt = pt - startPoint;
    const Vec3 edgeVec = endPoint - startPoint;
    const double normalizedDotProd = dot(c2pt, edgeVec) / (edgeLength * edgeLength);   // sqr(edgeLength)
    const Vec3 testProjPoint = startPoint + normalizedDotProd * edgeVec;
    if (normalizedDotProd <= ABS_TOL) {
        projPoint = startPoint;
        if (mag(testProjPoint - startPoint) <= distanceTolerance) edgePointI = startPointI;
    } else if (normalizedDotProd >= (1.0 - ABS_TOL)) {
        projPoint = endPoint;
        if (mag(testProjPoint - endPoint) <= distanceTolerance) edgePointI = endPointI;
    } else {
        projPoint = testProjPoint;
    }
}

// BPS.C:151-183; -1 = FatalError "Did not find any eligible corner points in edge mesh"
static int findClosestEdgeMeshCornerPointIndex(const Vec3& pt, const EdgeMesh& em) {
    double distance = GREAT;
    int closestPointI = UNDEF_LABEL;
    for (size_t pointI = 0; pointI < em.points.size(); ++pointI) {
        if (em.pointEdges[pointI].size() == 2) continue;
        const double testDistance = mag(pt - em.points[pointI]);
        if (testDistance < distance) { distance = testDistance; closestPointI = (int)pointI; }
    }
    return closestPointI;
}

// BPS.C:206-264; returns false on the reference's FatalError (no edge with the required string)
static bool findClosestEdgeInfo(const Vec3& pt, const EdgeMesh& em, int requiredStringI, const std::vector<int>& targetEdgeStrings,
                                double distanceTolerance, Vec3& projPoint, int& closestEdgeI, int& closestEdgeStringI, int& closestEdgePointI) {
    double distance = GREAT;
    projPoint = UNDEF_VECTOR;
    closestEdgeI = closestEdgeStringI = closestEdgePointI = UNDEF_LABEL;
    for (size_t edgeI = 0; edgeI < em.edges.size(); ++edgeI) {
        if ((requiredStringI >= 0) && (targetEdgeStrings[edgeI] != requiredStringI)) continue;
        Vec3 testProjPoint;
        int edgePointI;
        projectPointToEdge(pt, em, (int)edgeI, distanceTolerance, testProjPoint, edgePointI);
        const double testDistance = mag(testProjPoint - pt);
        if (testDistance < distance) {
            distance = testDistance;
            projPoint = testProjPoint;
            closestEdgeI = (int)edgeI;
            closestEdgePointI = edgePointI;
            if (em.edges.size() == targetEdgeStrings.size()) closestEdgeStringI = targetEdgeStrings[edgeI];
        }
    }
    return !((requiredStringI >= 0) && (closestEdgeStringI == UNDEF_LABEL));
}

// BPS.C:446-487
static void findContinuousEdgeMeshEdges(const EdgeMesh& em, int edgeI, int& neighEdgeI1, int& neighEdgeI2) {
    neighEdgeI1 = neighEdgeI2 = UNDEF_LABEL;
    const int pointI1 = em.edges[(size_t)edgeI][0];
    if (em.pointEdges[(size_t)pointI1].size() == 2) {
        int edgeI1 = em.pointEdges[(size_t)pointI1][0];
        if (edgeI1 == edgeI) edgeI1 = em.pointEdges[(size_t)pointI1][1];
        neighEdgeI1 = edgeI1;
    }
    const int pointI2 = em.edges[(size_t)edgeI][1];
    if (em.pointEdges[(size_t)pointI2].size() == 2) {
        int edgeI2 = em.pointEdges[(size_t)pointI2][0];
        if (edgeI2 == edgeI) edgeI2 = em.pointEdges[(size_t)pointI2][1];
        neighEdgeI2 = edgeI2;
    }
}

// BPS.C:492-551.  The recursion tests em.pointEdges()[<EDGE id>].size() == 2 (BPS.C:534,542): the list of the POINT
// that has the neighbour edge's id.  Kept as written; an id beyond the point list (undefined behaviour in the
// reference) counts as "not 2" here.
static void stringifyEdgeMeshEdges(const EdgeMesh& em, std::vector<int>& targetEdgeStrings, int edgeI, int neighEdgeI1, int neighEdgeI2, int& nStrings) {
    const int stringI0 = targetEdgeStrings[(size_t)edgeI];
    int stringI1 = UNDEF_LABEL;
    if (neighEdgeI1 != UNDEF_LABEL) stringI1 = targetEdgeStrings[(size_t)neighEdgeI1];
    int stringI2 = UNDEF_LABEL;
    if (neighEdgeI2 != UNDEF_LABEL) stringI2 = targetEdgeStrings[(size_t)neighEdgeI2];
    const int maxStringI = std::max(std::max(stringI0, stringI1), stringI2);
    if (maxStringI == UNDEF_LABEL) { ++nStrings; targetEdgeStrings[(size_t)edgeI] = nStrings; }
    else if (stringI0 == UNDEF_LABEL) targetEdgeStrings[(size_t)edgeI] = maxStringI;
    auto listOfTwo = [&](int id) { return (size_t)id < em.pointEdges.size() && em.pointEdges[(size_t)id].size() == 2; };
    if ((neighEdgeI1 != UNDEF_LABEL) && (stringI1 == UNDEF_LABEL) && listOfTwo(neighEdgeI1)) {
        int a, b;
        findContinuousEdgeMeshEdges(em, neighEdgeI1, a, b);
        stringifyEdgeMeshEdges(em, targetEdgeStrings, neighEdgeI1, a, b, nStrings);
    }
    if ((neighEdgeI2 != UNDEF_LABEL) && (stringI2 == UNDEF_LABEL) && listOfTwo(neighEdgeI2)) {
        int a, b;
        findContinuousEdgeMeshEdges(em, neighEdgeI2, a, b);
        stringifyEdgeMeshEdges(em, targetEdgeStrings, neighEdgeI2, a, b, nStrings);
    }
}

// BPS.C:557-587
int findEdgeMeshStrings(std::vector<int>& targetEdgeStrings, const EdgeMesh& em) {
    int nStrings = UNDEF_LABEL;
    targetEdgeStrings.assign(em.edges.size(), UNDEF_LABEL);
    for (size_t edgeI = 0; edgeI < em.edges.size(); ++edgeI) {
        if (targetEdgeStrings[edgeI] >= 0) continue;
        int a, b;
        findContinuousEdgeMeshEdges(em, (int)edgeI, a, b);
        stringifyEdgeMeshEdges(em, targetEdgeStrings, (int)edgeI, a, b, nStrings);
    }
    return nStrings;
}

// getMeshStats SM.C:1478-1541, the rank-local part: minimum edge length and bounding box of the edge end points
// (bb = min x, max x, min y, max y, min z, max z); the reference reduces them over the ranks (returnReduce :1528-1535)
void Domain::boundaryStatsLocal(double& minLength, double bb[6]) const {
    minLength = VGREAT;
    bb[0] = bb[2] = bb[4] = VGREAT;
    bb[1] = bb[3] = bb[5] = -VGREAT;
    for (const auto& e : edges) {
        const Vec3 startCoords = points[(size_t)e[0]], endCoords = points[(size_t)e[1]];
        const double length = mag(endCoords - startCoords);
        if (length < minLength) minLength = length;
        for (const Vec3& q : {startCoords, endCoords}) {
            if (q.x < bb[0]) bb[0] = q.x;
            if (q.y < bb[2]) bb[2] = q.y;
            if (q.z < bb[4]) bb[4] = q.z;
            if (q.x > bb[1]) bb[1] = q.x;
            if (q.y > bb[3]) bb[3] = q.y;
            if (q.z > bb[5]) bb[5] = q.z;
        }
    }
}

// serial set-up SM.C:2080-2253
void Domain::setupBoundary(const std::vector<Patch>& p, const LayerParams& lp, const BoundaryInput& in) {
    double minLength, bb[6];
    boundaryStatsLocal(minLength, bb);
    boundaryBegin(p, lp, in, minLength, bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4]);   // perimeter as SM.C:1538 ("+ bbMinZ")
    if (!error.empty()) return;
    const int maxIter = lay.maxLayers + 1;                                   // SM.C:2217
    for (int iter = 0; iter < maxIter; ++iter) layersHopsSweep();            // layer patches, OBB.C:83-131
    for (int iter = 0; iter < 2; ++iter) boundaryHopsSweep();                // smoothing patches, SM.C:2218
    calculateBoundaryPointNormals();                                         // SM.C:2219
    for (int iter = 1; iter < maxIter + 1; ++iter) layersPropagateSweep(iter);   // OBB.C:274-366
    layersUndo();
    boundaryFinish();
}

// Everything of the set-up that precedes the first synchronised step: storage, mesh statistics (already reduced over the
// ranks by the caller), prerequisites SM.C:2080-2093, edge mesh checks and strings SM.C:2131-2172, classifyBoundaryPoints
// BPS.C:269-441, zero hop counts on the layer and smoothing patches OBB.C:62-79.
void Domain::boundaryBegin(const std::vector<Patch>& p, const LayerParams& lp, const BoundaryInput& in, double minEdgeGlobal,
                           double perimeterGlobal) {
    layersBegin(p, lp);   // SM.C:1983-2028 storage, the layer part of classifyBoundaryPoints, zero hops of the layer patches
    if (!error.empty()) return;
    bnd = in;
    meshMinEdgeLength = minEdgeGlobal;
    meshPerimeter = perimeterGlobal;
    distanceTolerance = REL_TOL * std::min(meshMinEdgeLength, lay.layerEdgeLength);   // SM.C:1921

    bool labelIOListsHaveData = false;   // SM.C:2066-2077
    for (int v : bnd.isCornerPointIO) labelIOListsHaveData = labelIOListsHaveData || v == 1;
    for (int v : bnd.isFeatureEdgePointIO) labelIOListsHaveData = labelIOListsHaveData || v == 1;
    bool anySmoothingPatch = false;
    for (const Patch& pp : patches) anySmoothingPatch = anySmoothingPatch || pp.isSmoothingPatch;
    // SM.C:2080-2093 (file existence = non-empty input here)
    doBoundarySmoothing = !bnd.surf.tris.empty() && (!bnd.initEdges.edges.empty() || labelIOListsHaveData) && anySmoothingPatch;
    if (doBoundarySmoothing) {
        if (bnd.targetEdges.edges.empty()) bnd.targetEdges = bnd.initEdges;   // SM.C:2148-2160
        bnd.initEdges.buildPointEdges();
        bnd.targetEdges.buildPointEdges();
        error = checkEdgeMeshSanity(bnd.initEdges, meshMinEdgeLength, meshPerimeter);
        if (!error.empty()) return;
        error = checkEdgeMeshSanity(bnd.targetEdges, meshMinEdgeLength, meshPerimeter);
        if (!error.empty()) return;
        findEdgeMeshStrings(targetEdgeStrings, bnd.targetEdges);
    } else {
        bnd.initEdges = EdgeMesh();   // SM.C:2174-2180
        bnd.targetEdges = EdgeMesh();
    }
    const EdgeMesh& initEdges = bnd.initEdges;
    const EdgeMesh& targetEdges = bnd.targetEdges;

    isFeatureEdgePoint.assign(nPoints, 0);
    isCornerPoint.assign(nPoints, 0);
    isFrozenSurfacePoint.assign(nPoints, 0);
    isSmoothingSurfacePoint.assign(nPoints, 0);
    cornerPoints.assign(nPoints, UNDEF_VECTOR);
    pointStrings.assign(nPoints, UNDEF_LABEL);
    isCornerPointOut.assign(nPoints, 0);
    isFeatureEdgePointOut.assign(nPoints, 0);
    if ((int)bnd.isCornerPointIO.size() == nPoints) isCornerPointOut = bnd.isCornerPointIO;
    if ((int)bnd.isFeatureEdgePointIO.size() == nPoints) isFeatureEdgePointOut = bnd.isFeatureEdgePointIO;

    // classifyBoundaryPoints BPS.C:269-441 (isConnectedToInternalPoint / isLayerSurfacePoint were set by layersBegin in
    // the same visiting order)
    std::vector<unsigned char> isVisitedPoint(nPoints, 0);
    for (const Patch& pp : patches)
        for (int faceI = pp.start; faceI < pp.start + pp.size; ++faceI)
            for (int pointI : faces[faceI]) {
                if (isVisitedPoint[pointI]) continue;
                isVisitedPoint[pointI] = 1;
                if (isInternalPoint[pointI]) continue;
                if ((initEdges.points.size() > 0) && (targetEdges.points.size() > 0)) {
                    const Vec3 pt = points[pointI];
                    if (labelIOListsHaveData) {
                        isCornerPoint[pointI] = (isCornerPointOut[pointI] == 1) ? 1 : 0;
                        isFeatureEdgePoint[pointI] = (isFeatureEdgePointOut[pointI] == 1) ? 1 : 0;
                    } else {
                        Vec3 projPoint;
                        int dummy, dummy2, closestEdgePointI = UNDEF_LABEL;
                        findClosestEdgeInfo(pt, initEdges, -1, targetEdgeStrings, distanceTolerance, projPoint, dummy, dummy2, closestEdgePointI);
                        if ((closestEdgePointI >= 0) && (initEdges.pointEdges[(size_t)closestEdgePointI].size() != 2)) {
                            isCornerPoint[pointI] = 1;
                            isCornerPointOut[pointI] = 1;
                        } else if (mag(pt - projPoint) < distanceTolerance) {
                            isFeatureEdgePoint[pointI] = 1;
                            isFeatureEdgePointOut[pointI] = 1;
                        }
                    }
                    if (isCornerPoint[pointI]) {
                        const int closestCornerPointI = findClosestEdgeMeshCornerPointIndex(pt, targetEdges);
                        if (closestCornerPointI < 0) { error = "Did not find any eligible corner points in edge mesh"; return; }
                        cornerPoints[pointI] = targetEdges.points[(size_t)closestCornerPointI];
                    }
                }
                if (doBoundarySmoothing && pp.isSmoothingPatch) isSmoothingSurfacePoint[pointI] = 1;
                else isFrozenSurfacePoint[pointI] = 1;
            }

    // calculatePointHopsToBoundary(smoothingPatchIds, ..., 2) OBB.C:52-133: the zero hops; the two sweeps follow
    pointHopsToSmoothingBoundary.assign(nPoints, UNDEF_LABEL);
    for (const Patch& pp : patches) {
        if (!pp.isSmoothingPatch) continue;
        for (int faceI = pp.start; faceI < pp.start + pp.size; ++faceI)
            for (int patchPointI : faces[faceI])
                if (isConnectedToInternalPoint[patchPointI]) pointHopsToSmoothingBoundary[patchPointI] = 0;
    }
    smoothingNewHopCounts.assign(nPoints, -1);
}

// one sweep of calculatePointHopsToBoundary for the smoothing patches (OBB.C:85-121; the maxEq sync :124-130 follows)
void Domain::boundaryHopsSweep() {
    std::vector<int>& hops = pointHopsToSmoothingBoundary;
    std::vector<int>& newHopCounts = smoothingNewHopCounts;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (hops[pointI] >= 0) continue;
        if (!isInternalPoint[pointI]) continue;
        int maxHops = -1;
        for (int neighI : pointPoints[pointI])
            if (hops[neighI] > maxHops) maxHops = hops[neighI];
        if (maxHops >= 0) newHopCounts[pointI] = maxHops + 1;
    }
    for (int pointI = 0; pointI < nPoints; ++pointI)
        if (newHopCounts[pointI] > hops[pointI]) hops[pointI] = newHopCounts[pointI];
}

// propagateInnerNeighInfo OBB.C:396-459 and the target edge string of every feature edge point SM.C:2234-2249 (rank-local)
void Domain::boundaryFinish() {
    const EdgeMesh& targetEdges = bnd.targetEdges;
    isInnerNeighInProc.assign(nPoints, 0);
    pointToInnerPointMap.assign(nPoints, UNDEF_LABEL);
    innerNeighCoords.assign(nPoints, UNDEF_VECTOR);
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const int nHops = pointHopsToSmoothingBoundary[pointI];
        if (!isSmoothingSurfacePoint[pointI]) continue;
        if (!isConnectedToInternalPoint[pointI]) continue;
        if (nHops != 0) { error = std::to_string(pointI) + " is not boundary point"; return; }
        int nNeighHops = 0, neighPointI = UNDEF_LABEL;
        for (int neighI : pointPoints[pointI])
            if (pointHopsToSmoothingBoundary[neighI] == (nHops + 1)) { ++nNeighHops; neighPointI = neighI; }
        if (nNeighHops == 1) { isInnerNeighInProc[pointI] = 1; pointToInnerPointMap[pointI] = neighPointI; }
    }
    if (doBoundarySmoothing)
        for (int pointI = 0; pointI < nPoints; ++pointI) {
            if (!isFeatureEdgePoint[pointI]) continue;
            Vec3 dummyPoint;
            int dummy, dummy2, pointStringI = UNDEF_LABEL;
            findClosestEdgeInfo(points[pointI], targetEdges, -1, targetEdgeStrings, distanceTolerance, dummyPoint, dummy, pointStringI, dummy2);
            pointStrings[pointI] = pointStringI;
        }
}

// OpenFOAM triangle::intersection(orig, dir, intersection::HALF_RAY, tol)
static bool triangleIntersection(const Vec3& a, const Vec3& b, const Vec3& c, const Vec3& orig, const Vec3& dir, double tol, double& t, Vec3& pt) {
    const Vec3 edge1 = b - a;
    const Vec3 edge2 = c - a;
    const Vec3 pVec = cross(dir, edge2);
    const double det = dot(edge1, pVec);
    if (det > -ROOTVSMALL && det < ROOTVSMALL) return false;
    const double inv_det = 1.0 / det;
    const Vec3 tVec = orig - a;
    const double u = dot(tVec, pVec) * inv_det;
    if (u < -tol || u > 1.0 + tol) return false;
    const Vec3 qVec = cross(tVec, edge1);
    const double v = dot(dir, qVec) * inv_det;
    if (v < -tol || u + v > 1.0 + tol) return false;
    t = dot(edge2, qVec) * inv_det;
    if (t < -tol) return false;
    pt = a + u * edge1 + v * edge2;
    return true;
}

Vec3 Domain::findLine(const Vec3& start, const Vec3& end, bool& hit) const {
    const Vec3 dir = end - start;
    const double tol = 10.0 * SMALL;   // indexedOctree::perturbTol()
    double best = 0.0;
    Vec3 bestPt = UNDEF_VECTOR;
    hit = false;
    for (const auto& tr : bnd.surf.tris) {
        double t;
        Vec3 pt;
        if (!triangleIntersection(bnd.surf.points[(size_t)tr[0]], bnd.surf.points[(size_t)tr[1]], bnd.surf.points[(size_t)tr[2]], start, dir, tol, t, pt)) continue;
        if (!(t <= 1.0)) continue;   // treeDataTriSurface::findIntersectOp: inter.distance() <= 1
        if (!hit || t < best) { hit = true; best = t; bestPt = pt; }
    }
    return bestPt;
}

// BPS.C:682-745
Vec3 Domain::findIntersection(const Vec3& origPoint, const Vec3& pointNormal, double searchDistance) const {
    Vec3 hitPoint1 = UNDEF_VECTOR;
    {
        const Vec3 endPoint = origPoint + pointNormal * searchDistance;
        bool hit;
        const Vec3 h = findLine(origPoint, endPoint, hit);
        if (hit) hitPoint1 = h;
    }
    Vec3 hitPoint2 = UNDEF_VECTOR;
    {
        const Vec3 endPoint = origPoint - pointNormal * searchDistance;
        bool hit;
        const Vec3 h = findLine(origPoint, endPoint, hit);
        if (hit) hitPoint2 = h;
    }
    const double distance1 = mag(origPoint - hitPoint1);
    const double distance2 = mag(origPoint - hitPoint2);
    if (distance1 < distance2) return hitPoint1;
    else if (distance2 < distance1) return hitPoint2;
    {
        const Vec3 endPoint1 = origPoint + pointNormal * searchDistance;
        const Vec3 endPoint2 = origPoint - pointNormal * searchDistance;
        bool hit;
        const Vec3 h = findLine(endPoint1, endPoint2, hit);
        if (hit) return h;
    }
    return UNDEF_VECTOR;
}

// The parts of SM.C:2307-2330 that only read the current coordinates, before their syncPointList calls: the local half of
// updateNeighCoords with the inner maps (OBB.C:471-486; minMagSqr sync :490-496) and of calculateFeatureEdgeProjections
// (BPS.C:637-656; plusEq syncs :659-674).
void Domain::boundaryLocalPre() {
    const EdgeMesh& targetEdges = bnd.targetEdges;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (!isInnerNeighInProc[pointI]) { innerNeighCoords[pointI] = UNDEF_VECTOR; continue; }
        innerNeighCoords[pointI] = points[(size_t)pointToInnerPointMap[pointI]];
    }
    featureEdgeProjections.assign(nPoints, ZERO_VECTOR);
    nFeatureEdgeProjections.assign(nPoints, 0);
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (!isFeatureEdgePoint[pointI]) continue;
        for (int neighI : pointPoints[pointI]) {   // findNeighborSurfacePoints BPS.C:592-615
            if (isInternalPoint[neighI]) continue;
            if (isFeatureEdgePoint[neighI]) continue;
            if (isCornerPoint[neighI]) continue;
            Vec3 projPoint;
            int d1, d2, d3;
            if (!findClosestEdgeInfo(points[neighI], targetEdges, pointStrings[pointI], targetEdgeStrings, distanceTolerance, projPoint, d1, d2, d3)) {
                error = "Internal sanity check failed: Did not find any edges with string index";
                return;
            }
            featureEdgeProjections[pointI] += projPoint;
            ++nFeatureEdgeProjections[pointI];
        }
    }
}

// SM.C:2307-2357 after the synchronised inputs are in place: projectBoundaryPointsToEdgesAndSurfaces BPS.C:843-945,
// projectPrismaticInternalPointsToSurfaces OBB.C:573-631, constrainMaxStepLength over all points
void Domain::projectBoundaryPoints() {
    // calculateSurfaceCentroids BPS.C:781-839: the centroids are blended with faceCentroidBlendingFraction = 0.0
    // (BPS.C:872): the term is 0 * centroid, which changes no finite value; not restated
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (isInternalPoint[pointI]) continue;
        if (isCornerPoint[pointI]) { newPoints[pointI] = cornerPoints[pointI]; continue; }
        if (isFeatureEdgePoint[pointI]) {
            newPoints[pointI] = featureEdgeProjections[pointI] / double(nFeatureEdgeProjections[pointI]);
            continue;
        }
        if (isSharpEdgePoint[pointI]) isFrozenPoint[pointI] = 1;
        else if (isSmoothingSurfacePoint[pointI]) {
            const Vec3 pointNormal = pointNormals[pointI];
            if (pointNormal == ZERO_VECTOR) { error = "pointNormal is zero for pointI " + std::to_string(pointI); return; }
            double searchDistance = distanceTolerance;
            const Vec3 newPoint = newPoints[pointI];   // 0 * centroid + (1 - 0) * newPoints
            Vec3 surfPoint = UNDEF_VECTOR;
            for (int i = 0; i < 4; ++i) {
                searchDistance *= (1.0 / REL_TOL);
                surfPoint = findIntersection(newPoint, pointNormal, searchDistance);
                if (surfPoint != UNDEF_VECTOR) { newPoints[pointI] = surfPoint; break; }
            }
            if (surfPoint == UNDEF_VECTOR) { error = "Did not find surface intersection for pointI " + std::to_string(pointI); return; }
        }
    }
    // projectPrismaticInternalPointsToSurfaces OBB.C:573-631
    const double f = bnd.internalSmoothingBlendingFraction;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (!isSmoothingSurfacePoint[pointI]) continue;
        if (!isConnectedToInternalPoint[pointI]) continue;
        if (pointToInnerPointMap[pointI] < 0) continue;
        if (isFeatureEdgePoint[pointI]) continue;
        if (isCornerPoint[pointI]) continue;
        if (isSharpEdgePoint[pointI]) continue;
        const Vec3 pointNormal = pointNormals[pointI];
        const Vec3 innerNeighCoord = innerNeighCoords[pointI];
        if (pointNormal == ZERO_VECTOR) { error = "Point has zero point normal"; return; }
        if (innerNeighCoord == UNDEF_VECTOR) { error = "Point has no inner neigh coord"; return; }
        const Vec3 cCoords = newPoints[pointI];
        const Vec3 neighVec = cCoords - innerNeighCoord;
        const double dotProd = dot(neighVec, pointNormal);
        const Vec3 pVec = neighVec - dotProd * pointNormal;
        const Vec3 newCoords = cCoords - pVec;
        newPoints[pointI] = f * newCoords + (1 - f) * newPoints[pointI];
    }
    // SM.C:2356 constrainMaxStepLength
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const Vec3 cCoords = points[pointI];
        const Vec3 stepDir = newPoints[pointI] - cCoords;
        double globalScale;
        if (mag(stepDir) > prm.maxStepLength) globalScale = prm.maxStepLength / (mag(stepDir) * prm.relStepFrac);
        else globalScale = 1.0;
        newPoints[pointI] = cCoords + (prm.relStepFrac * globalScale) * stepDir;
    }
}

}  // namespace orc
