// smooth_oracle.cpp -- CPU restatement of the reference's smoothing iteration loop.
// TEST INFRASTRUCTURE ONLY (see smooth_oracle.hpp).  PARITY UNPINNED (see header).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off  (no FMA contraction, no fast-math: the
// x86-64 reference build evaluates every expression in plain IEEE f64, left to right).
//
// Every function cites the reference lines it follows (SM.C = src/smoothMesh.C).
#include "smooth_oracle.hpp"
#include "oracle_vec.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <future>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <cstring>
#include <stack>

#include "../smoothmesh_amd/csrc/smacos.hpp"      // the product's acos as a fixed sequence of IEEE operations (host path: plain C++)

namespace orc {

// acos variant: 0 = glibc's std::acos, the reference's arithmetic (default); 1 = the algorithm the device kernels evaluate
// (smacos::acosX, bit-identical on CPU and GPU) -- with it the engine's angle fields can be compared with ANGLE_TOL = 0.
// Process-wide: edgeEdgeAngle / calcEdgeCenterEdgeAngle are free functions, as in the reference.
static int g_acosVariant = 0;
void setAcosVariant(int v) { g_acosVariant = v; }
int acosVariant() { return g_acosVariant; }
static inline double acosV(double c) { return g_acosVariant ? smacos::acosX(c) : std::acos(c); }

// census of the threshold comparisons an angle takes part in (SM.C:923, 1367, 1391-1394, 1421-1424): how many there were and how
// many had their two sides within 8 ulp of each other -- the comparisons a last-bit difference between two acos implementations
// could flip.  Counted while enabled (tests / bench oracle legs); not thread-safe, like the loop itself.
static AcosCensus g_census;
static bool g_censusOn = false;
void censusEnable(bool on) { g_censusOn = on; }
void censusReset() { g_census = AcosCensus(); }
AcosCensus censusGet() { return g_census; }
static unsigned long long g_censusWindow = 4ull;
void censusWindow(unsigned long long ulps) { g_censusWindow = ulps; }
// cls < 0: counted in the totals only (a re-visit of a point the engine's census sees once per iteration)
static inline void censusNote(double a, double b, int cls) {
    if (!g_censusOn) return;
    ++g_census.comparisons;
    if (!(a == a) || !(b == b)) return;
    long long ia, ib;
    std::memcpy(&ia, &a, 8); std::memcpy(&ib, &b, 8);
    if (ia < 0) ia = (long long)0x8000000000000000ull - ia;      // (monotone map of the doubles onto the integers)
    if (ib < 0) ib = (long long)0x8000000000000000ull - ib;
    const unsigned long long d = ia > ib ? (unsigned long long)ia - (unsigned long long)ib : (unsigned long long)ib - (unsigned long long)ia;
    // (equal sides: the same function of the same inputs on both sides -- a point that does not move, an angle that is its own
    // minimum -- which no implementation of acos can tell apart; counted on their own)
    if (d == 0ull) { ++g_census.equal; return; }
    if (d <= 8ull) ++g_census.within8ulp;
    if (cls >= 0 && d <= g_censusWindow) ++g_census.near[cls];
    if (d < g_census.minUlp) g_census.minUlp = d;
}
static inline bool lessC(double a, double b, int cls) { censusNote(a, b, cls); return a < b; }
static inline bool greaterC(double a, double b, int cls) { censusNote(a, b, cls); return a > b; }

// SM.C:172-180
static inline double getPointDistance(const Vec3& coords1, const Vec3& coords2) {
    const Vec3 v = coords2 - coords1;
    return mag(v);
}

// SM.C:766-786
double edgeEdgeAngle(const Vec3& cCoords, const Vec3& p1Coords, const Vec3& p2Coords) {
    Vec3 vec1 = (p1Coords - cCoords);
    Vec3 vec2 = (p2Coords - cCoords);
    vec1 /= mag(vec1);
    vec2 /= mag(vec2);
    const double cosA = dot(vec1, vec2);
    const double MAX = 0.99999;
    // std::max/std::min argument order kept: a NaN cosA maps to +MAX
    const double cosAlpha = std::max(-MAX, std::min(MAX, cosA));
    return acosV(cosAlpha);
}

// SM.C:980-998
double calcEdgeCenterEdgeAngle(const Vec3& p0, const Vec3& cC, const Vec3& p1) {
    const double cosA0 = dot(p0, cC);
    const double cosA1 = dot(cC, p1);
    const double MAX = 0.99999;
    const double cosAlpha0 = std::max(-MAX, std::min(MAX, cosA0));
    const double angle0 = acosV(cosAlpha0);
    const double cosAlpha1 = std::max(-MAX, std::min(MAX, cosA1));
    const double angle1 = acosV(cosAlpha1);
    return angle0 + angle1;
}

// SM.C:222-239
static bool isSmallerByVectorElements(const Vec3& v1, const Vec3& v2) {
    const double a[3] = {v1.x, v1.y, v1.z}, b[3] = {v2.x, v2.y, v2.z};
    for (int i = 0; i < 3; ++i) {
        if (a[i] < b[i]) return true;
        else if (a[i] > b[i]) return false;
    }
    return false;
}

// SM.C:246-272
bool isCloserPoint(const Vec3& point1, const Vec3& point2) {
    if (point1 == point2) return false;
    const double deltaDistance = mag(point1) - mag(point2);
    if (deltaDistance < VSMALL) return true;
    else if ((std::abs(deltaDistance) < VSMALL) && isSmallerByVectorElements(point1, point2)) return true;
    return false;
}

// ---- addressing -----------------------------------------------------------------------
static int findIndex(const std::vector<int>& l, int v) {
    for (size_t i = 0; i < l.size(); ++i)
        if (l[i] == v) return int(i);
    return -1;
}

void Domain::build() {
    error.clear();
    const bool verbose = std::getenv("ORACLE_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (verbose) std::fprintf(stderr, "[oracle] build: %-18s done at %.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    nFaces = int(faces.size());
    nInternalFaces = int(neighbour.size());

    // The tables below are three independent chains over (faces, owner, neighbour); each is built exactly as before,
    // sequentially in itself, on a host thread of its own (set-up time only: 10 M-cell meshes took minutes).
    auto chainCells = [&] {
        // OpenFOAM primitiveMesh::calcCells: owned faces ascending, then neighboured faces ascending
        std::vector<std::vector<int>> cells(nCells);
        for (int f = 0; f < nFaces; ++f) cells[owner[f]].push_back(f);
        for (int f = 0; f < nInternalFaces; ++f) cells[neighbour[f]].push_back(f);
        // cellPoints: cell::labels(faces) = first-appearance order walking the cell's faces
        cellPoints.assign(nCells, {});
        for (int c = 0; c < nCells; ++c)
            for (int f : cells[c])
                for (int p : faces[f])
                    if (findIndex(cellPoints[c], p) < 0) cellPoints[c].push_back(p);
        lap("cellPoints");
        // pointCells: OpenFOAM calcPointCells walks cells ascending -> ascending cell id
        pointCells.assign(nPoints, {});
        for (int c = 0; c < nCells; ++c)
            for (int p : cellPoints[c]) pointCells[p].push_back(c);
        lap("pointCells");
        // SM.C:190-217 generatePointNeighPoints
        pointNeighPoints.assign(nPoints, {});
        for (int p = 0; p < nPoints; ++p)
            for (int c : pointCells[p])
                for (int q : cellPoints[c]) {
                    if (p == q) continue;
                    if (findIndex(pointNeighPoints[p], q) == -1) pointNeighPoints[p].push_back(q);
                }
        lap("pointNeighPoints");
    };
    auto chainFaces = [&] {
        // pointFaces: invertManyToMany(nPoints, faces) -> ascending face id
        pointFaces.assign(nPoints, {});
        for (int f = 0; f < nFaces; ++f)
            for (int p : faces[f]) pointFaces[p].push_back(f);
        lap("pointFaces");
        // SM.C:1575-1620 generateCellFaces: internal faces by owner, internal faces by neighbour,
        // then boundary faces patch by patch (= ascending face id)
        cellFaces.assign(nCells, {});
        for (int f = 0; f < nInternalFaces; ++f) cellFaces[owner[f]].push_back(f);
        for (int f = 0; f < nInternalFaces; ++f) cellFaces[neighbour[f]].push_back(f);
        for (int f = nInternalFaces; f < nFaces; ++f) cellFaces[owner[f]].push_back(f);
        lap("cellFaces");
    };
    auto chainEdges = [&] {
        // edges: OpenFOAM primitiveMesh::calcEdges, unsorted-points branch (nInternalPoints_ == -1):
        // edges stored (min,max) and renumbered into upper-triangular order (start ascending,
        // then end ascending); pointEdges sorted ascending; pointPoints[p][i] = other end of
        // pointEdges[p][i]  (=> ascending neighbour point id).
        // (built as a sorted list of (min, max) keys rather than an ordered map: the same order -- lexicographic in
        // (start, end) -- at a fraction of the set-up time; edgeStart[a] = first edge starting at a)
        std::vector<uint64_t> keys;
        {
            size_t nnz = 0;
            for (int f = 0; f < nFaces; ++f) nnz += faces[f].size();
            keys.reserve(nnz);
        }
        for (int f = 0; f < nFaces; ++f) {
            const auto& fp = faces[f];
            const int n = int(fp.size());
            for (int i = 0; i < n; ++i) {
                const int a = fp[i], b = fp[(i + 1) % n];
                keys.push_back((uint64_t(uint32_t(std::min(a, b))) << 32) | uint32_t(std::max(a, b)));
            }
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        edges.clear();
        edges.reserve(keys.size());
        std::vector<int> edgeStart(size_t(nPoints) + 1, 0);
        for (uint64_t k : keys) {
            edges.push_back({int(k >> 32), int(k & 0xffffffffu)});
            ++edgeStart[size_t(k >> 32) + 1];
        }
        for (int p = 0; p < nPoints; ++p) edgeStart[size_t(p) + 1] += edgeStart[size_t(p)];
        { std::vector<uint64_t>().swap(keys); }
        auto findEdge = [&](int a, int b) -> int {
            const int lo = std::min(a, b), hi = std::max(a, b);
            for (int e = edgeStart[size_t(lo)]; e < edgeStart[size_t(lo) + 1]; ++e)
                if (edges[e][1] == hi) return e;
            return -1;
        };
        const int nEdges = int(edges.size());
        pointEdges.assign(nPoints, {});
        for (int e = 0; e < nEdges; ++e) {
            pointEdges[edges[e][0]].push_back(e);
            pointEdges[edges[e][1]].push_back(e);
        }
        pointPoints.assign(nPoints, {});
        for (int p = 0; p < nPoints; ++p)
            for (int e : pointEdges[p]) pointPoints[p].push_back(edges[e][0] == p ? edges[e][1] : edges[e][0]);
        lap("edges/pointEdges");
        // edgeFaces: ascending face id
        edgeFaces.assign(nEdges, {});
        for (int f = 0; f < nFaces; ++f) {
            const auto& fp = faces[f];
            const int n = int(fp.size());
            for (int i = 0; i < n; ++i) edgeFaces[findEdge(fp[i], fp[(i + 1) % n])].push_back(f);
        }
        lap("edgeFaces");
        // edgeCells: primitiveMesh::edgeCells(edgeI, storage) on-the-fly form = first appearance
        // walking edgeFaces (owner then neighbour).  Only min/max reductions consume it.
        edgeCells.assign(nEdges, {});
        for (int e = 0; e < nEdges; ++e)
            for (int f : edgeFaces[e]) {
                if (findIndex(edgeCells[e], owner[f]) < 0) edgeCells[e].push_back(owner[f]);
                if (f < nInternalFaces && findIndex(edgeCells[e], neighbour[f]) < 0) edgeCells[e].push_back(neighbour[f]);
            }
        lap("edgeCells");
    };
    if (nFaces < 200000 || std::getenv("ORACLE_SERIAL_BUILD")) {
        chainCells(); chainFaces(); chainEdges();
    } else {
        auto f1 = std::async(std::launch::async, chainCells);
        auto f2 = std::async(std::launch::async, chainFaces);
        chainEdges();
        f1.get(); f2.get();
    }
    isFrozenPoint.assign(nPoints, 0);
}

// SM.C:1478-1541 (edge length part)
void Domain::meshStats(double& minEdge, double& maxEdge) const {
    double minLength = VGREAT, maxLength = 0.0;
    for (const auto& e : edges) {
        const Vec3 startCoords = points[e[0]];
        const Vec3 endCoords = points[e[1]];
        const double length = mag(endCoords - startCoords);
        if (length < minLength) minLength = length;
        if (length > maxLength) maxLength = length;
    }
    minEdge = minLength;
    maxEdge = maxLength;
}

// ---- OpenFOAM primitiveMesh geometry (triggered by mesh.cellCentres(), SM.C:129) -------
void Domain::updateGeometry() {
    const std::vector<Vec3>& p = points;
    faceCentres.resize(nFaces);
    faceAreas.resize(nFaces);
    // primitiveMesh::makeFaceCentresAndAreas (.com v2412)
    for (int facei = 0; facei < nFaces; ++facei) {
        const std::vector<int>& f = faces[facei];
        const int nP = int(f.size());
        if (nP == 3) {
            faceCentres[facei] = (1.0 / 3.0) * (p[f[0]] + p[f[1]] + p[f[2]]);
            faceAreas[facei] = 0.5 * cross(p[f[1]] - p[f[0]], p[f[2]] - p[f[0]]);
        } else if (foamVariant == 1) {
            // OpenFOAM.org 12, primitiveMeshFaceCentresAndAreas.C (face::centre / face::area): the triangle areas are
            // projected on the face normal, so that the centre does not depend on the point average it is built around
            Vec3 pAvg = p[f[0]];
            for (int pi = 1; pi < nP; ++pi) pAvg += p[f[pi]];
            pAvg /= double(nP);
            Vec3 sumA = ZERO_VECTOR;
            for (int pi = 0; pi < nP; ++pi) {
                const Vec3 thisPoint = p[f[pi]], nextPoint = p[f[pi == nP - 1 ? 0 : pi + 1]];
                sumA += cross(nextPoint - thisPoint, pAvg - thisPoint);
            }
            const Vec3 sumAHat = sumA / mag(sumA);   // normalised(sumA)
            double sumAn = 0.0;
            Vec3 sumAnc = ZERO_VECTOR;
            for (int pi = 0; pi < nP; ++pi) {
                const Vec3 thisPoint = p[f[pi]], nextPoint = p[f[pi == nP - 1 ? 0 : pi + 1]];
                const Vec3 a = cross(nextPoint - thisPoint, pAvg - thisPoint);
                const Vec3 c = thisPoint + nextPoint + pAvg;
                const double an = dot(a, sumAHat);
                sumAn += an;
                sumAnc += an * c;
            }
            if (sumAn > VSMALL) faceCentres[facei] = ((1.0 / 3.0) * sumAnc) / sumAn;
            else faceCentres[facei] = pAvg;
            faceAreas[facei] = 0.5 * sumA;
        } else {
            Vec3 sumN = ZERO_VECTOR;
            double sumA = 0.0;
            Vec3 sumAc = ZERO_VECTOR;
            Vec3 fCentre = p[f[0]];
            for (int pi = 1; pi < nP; ++pi) fCentre += p[f[pi]];
            fCentre /= double(nP);
            for (int pi = 0; pi < nP; ++pi) {
                const int nextPi = (pi == nP - 1 ? 0 : pi + 1);
                const Vec3 nextPoint = p[f[nextPi]];
                const Vec3 thisPoint = p[f[pi]];
                const Vec3 c = thisPoint + nextPoint + fCentre;
                const Vec3 n = cross(nextPoint - thisPoint, fCentre - thisPoint);
                const double a = mag(n);
                sumN += n;
                sumA += a;
                sumAc += a * c;
            }
            if (sumA < ROOTVSMALL) {
                faceCentres[facei] = fCentre;
                faceAreas[facei] = ZERO_VECTOR;
            } else {
                faceCentres[facei] = ((1.0 / 3.0) * sumAc) / sumA;
                faceAreas[facei] = 0.5 * sumN;
            }
        }
    }
    // primitiveMesh::makeCellCentresAndVols (.com v2412)
    std::vector<Vec3> cEst(nCells, ZERO_VECTOR);
    std::vector<int> nCellFaces(nCells, 0);
    cellCentres.assign(nCells, ZERO_VECTOR);
    std::vector<double> cellVols(nCells, 0.0);
    for (int facei = 0; facei < nFaces; ++facei) {
        cEst[owner[facei]] += faceCentres[facei];
        ++nCellFaces[owner[facei]];
    }
    for (int facei = 0; facei < nInternalFaces; ++facei) {
        cEst[neighbour[facei]] += faceCentres[facei];
        ++nCellFaces[neighbour[facei]];
    }
    for (int celli = 0; celli < nCells; ++celli) cEst[celli] /= double(nCellFaces[celli]);
    for (int facei = 0; facei < nFaces; ++facei) {
        const Vec3 fc = faceCentres[facei];
        const Vec3 fA = faceAreas[facei];
        double pyr3Vol = dot(fA, fc - cEst[owner[facei]]);
        if (foamVariant == 1) pyr3Vol = (pyr3Vol > VSMALL) ? pyr3Vol : VSMALL;   // OpenFOAM.org: max(Sf & (Cf - cEst), vSmall), Foam::max(a, b) = (a > b) ? a : b
        const Vec3 pc = (3.0 / 4.0) * fc + (1.0 / 4.0) * cEst[owner[facei]];
        cellCentres[owner[facei]] += pyr3Vol * pc;
        cellVols[owner[facei]] += pyr3Vol;
    }
    for (int facei = 0; facei < nInternalFaces; ++facei) {
        const Vec3 fc = faceCentres[facei];
        const Vec3 fA = faceAreas[facei];
        double pyr3Vol = dot(fA, cEst[neighbour[facei]] - fc);
        if (foamVariant == 1) pyr3Vol = (pyr3Vol > VSMALL) ? pyr3Vol : VSMALL;
        const Vec3 pc = (3.0 / 4.0) * fc + (1.0 / 4.0) * cEst[neighbour[facei]];
        cellCentres[neighbour[facei]] += pyr3Vol * pc;
        cellVols[neighbour[facei]] += pyr3Vol;
    }
    for (int celli = 0; celli < nCells; ++celli) {
        if (std::abs(cellVols[celli]) > VSMALL) cellCentres[celli] /= cellVols[celli];
        else cellCentres[celli] = cEst[celli];
    }
}

// SM.C:277-308
static int findAppropriateClosestPointLabel(const std::vector<int>& pointPoints, const std::vector<int>& sLabels,
                                            int pointI, const std::vector<unsigned char>& isInternalPoint,
                                            int stride) {
    const bool isThisInternalPoint = isInternalPoint[pointI];
    int counter = 0;
    for (size_t i = 0; i < sLabels.size(); ++i) {
        const int labelI = sLabels[i];
        if ((!isThisInternalPoint) && (isInternalPoint[pointPoints[labelI]])) continue;
        if (counter == stride) return labelI;
        ++counter;
    }
    return -1;  // UNDEF_LABEL
}

// SM.C:489-543
static double calcARSmoothingRatio(const Vec3& closestPoint1, const Vec3& closestPoint2, const Vec3& closestPoint3,
                                   bool hasCommonCell, bool isInternalPoint) {
    if (hasCommonCell) return 0.0;
    if ((closestPoint1 == ZERO_VECTOR) || (closestPoint2 == ZERO_VECTOR)) return 0.0;
    const double lengthRatio1 = mag(closestPoint2) / mag(closestPoint1);
    const double lengthRatio2 = mag(closestPoint3) / mag(closestPoint2);
    if (isInternalPoint) {
        const double minRatio = 1.5;
        const double maxRatio = 3.0;
        if ((lengthRatio1 < minRatio) && (lengthRatio2 > minRatio)) {
            const double frac = (lengthRatio2 - minRatio) / (maxRatio - minRatio);
            const double blendFrac = std::min(1.0, std::max(0.0, frac));
            return blendFrac;
        }
    } else {
        const double minRatio = 1.0;
        const double maxRatio = 2.0;
        const double frac = (lengthRatio1 - minRatio) / (maxRatio - minRatio);
        const double blendFrac = std::min(1.0, std::max(0.0, frac));
        return blendFrac;
    }
    return 0.0;
}

// ---- boundary layer treatment: setup (serial) ---------------------------------------------------------

// OBB.C:141-233.  Note the reference never resets pointNormals: a boundary point's new normal is the
// normalised sum of its previous (unit) normal and the inverted unit normals of its boundary faces, and every
// non-zero normal (also the ones copied to internal points) is divided by its magnitude again on every call.
void Domain::layersNormalsAccumulate() {
    layerNFaces.assign(nPoints, 0);
    for (const Patch& pp : patches) {
        if (pp.kind == 1) continue;   // processor
        if (pp.kind == 2) continue;   // empty
        for (int faceI = 0; faceI < pp.size; ++faceI) {
            const Vec3 cSf = faceAreas[pp.start + faceI];          // fvPatch::Sf
            const Vec3 Sf = cSf / mag(cSf);                        // / magSf (= mag(Sf))
            for (int pointI : faces[pp.start + faceI]) {
                pointNormals[pointI] -= Sf;
                ++layerNFaces[pointI];
            }
        }
    }
}
void Domain::layersNormalsFinish() {
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (layerNFaces[pointI] < 1) continue;
        const double magNorm = mag(pointNormals[pointI]);
        if (magNorm < 0.1) { pointNormals[pointI] = ZERO_VECTOR; isSharpEdgePoint[pointI] = 1; }
        else isSharpEdgePoint[pointI] = 0;
    }
    for (int pointI = 0; pointI < nPoints; ++pointI)
        if (pointNormals[pointI] != ZERO_VECTOR) pointNormals[pointI] /= mag(pointNormals[pointI]);
}
void Domain::calculateBoundaryPointNormals() {
    layersNormalsAccumulate();
    layersNormalsFinish();
}

void Domain::layersBegin(const std::vector<Patch>& p, const LayerParams& lp) {
    patches = p;
    lay = lp;
    bool anyLayer = false;
    for (const Patch& pp : patches) anyLayer = anyLayer || pp.isLayerPatch;
    doLayerTreatment = anyLayer && (lay.layerMaxBlendingFraction > SMALL);   // SM.C:2024-2028
    pointHopsToLayerBoundary.assign(nPoints, -1);      // UNDEF_LABEL, SM.C:1983
    pointNormals.assign(nPoints, ZERO_VECTOR);         // SM.C:1987
    outerNeighCoords.assign(nPoints, UNDEF_VECTOR);    // SM.C:1993
    isOuterNeighInProc.assign(nPoints, 0);
    pointToOuterPointMap.assign(nPoints, -1);
    isConnectedToInternalPoint.assign(nPoints, 0);
    isLayerSurfacePoint.assign(nPoints, 0);
    isSharpEdgePoint.assign(nPoints, 0);
    layerNewHopCounts.assign(nPoints, -1);
    layerNFaces.assign(nPoints, 0);
    // boundaryPointLabels[q] = outer neighbour of q (OBB.C:258); the reference looks a label up with findIndex
    // (lowest q holding it).  All points that map to the same neighbour have the same hop count, so they are met
    // in one sweep in ascending order and the first one recorded IS the lowest: firstMapper replaces the O(P) scan.
    layerFirstMapper.assign(nPoints, -1);
    updateGeometry();

    // classifyBoundaryPoints BPS.C:296-340, 397-403: every point is classified by the first patch it is met on
    std::vector<unsigned char> isVisitedPoint(nPoints, 0);
    for (const Patch& pp : patches)
        for (int faceI = pp.start; faceI < pp.start + pp.size; ++faceI)
            for (int pointI : faces[faceI]) {
                if (isVisitedPoint[pointI]) continue;
                isVisitedPoint[pointI] = 1;
                if (isInternalPoint[pointI]) continue;
                for (int i : pointPoints[pointI])
                    if (isInternalPoint[i]) isConnectedToInternalPoint[pointI] = 1;
                if (pp.isLayerPatch) isLayerSurfacePoint[pointI] = 1;
            }
    // calculatePointHopsToBoundary OBB.C:62-79: zero hops on the layer patches
    for (const Patch& pp : patches) {
        if (!pp.isLayerPatch) continue;
        for (int faceI = pp.start; faceI < pp.start + pp.size; ++faceI)   // getPatchPointIndices OBB.C:22-46
            for (int patchPointI : faces[faceI])
                if (isConnectedToInternalPoint[patchPointI]) pointHopsToLayerBoundary[patchPointI] = 0;
    }
}

void Domain::layersHopsSweep() {
    std::vector<int>& hops = pointHopsToLayerBoundary;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (hops[pointI] >= 0) continue;
        if (!isInternalPoint[pointI]) continue;
        int maxHops = -1;
        for (int neighI : pointPoints[pointI])
            if (hops[neighI] > maxHops) maxHops = hops[neighI];
        if (maxHops >= 0) layerNewHopCounts[pointI] = maxHops + 1;
    }
    for (int pointI = 0; pointI < nPoints; ++pointI)
        if (layerNewHopCounts[pointI] > hops[pointI]) hops[pointI] = layerNewHopCounts[pointI];
}

void Domain::layersPropagateSweep(int iter) {
    const std::vector<int>& hops = pointHopsToLayerBoundary;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const int nHops = hops[pointI];
        if (nHops != iter) continue;
        int nNeighHops = 0;
        int neighPointI = -1;
        for (int neighI : pointPoints[pointI])
            if (hops[neighI] == (nHops - 1)) { ++nNeighHops; neighPointI = neighI; }
        if (nNeighHops == 1) {
            if ((!isInternalPoint[neighPointI]) && (!isLayerSurfacePoint[neighPointI])) continue;
            const int prevPointI = layerFirstMapper[neighPointI];   // findIndex(boundaryPointLabels, neighPointI)
            if (prevPointI >= 0) {
                pointNormals[pointI] = UNDEF_VECTOR;
                pointNormals[prevPointI] = UNDEF_VECTOR;
                continue;
            }
            isOuterNeighInProc[pointI] = 1;
            pointToOuterPointMap[pointI] = neighPointI;
            pointNormals[pointI] = pointNormals[neighPointI];
            layerFirstMapper[neighPointI] = pointI;
        }
    }
}

void Domain::layersUndo() {
    for (int pointI = 0; pointI < nPoints; ++pointI)
        if (pointNormals[pointI] == UNDEF_VECTOR) {
            pointNormals[pointI] = ZERO_VECTOR;
            isOuterNeighInProc[pointI] = 0;
            pointToOuterPointMap[pointI] = -1;
        }
}

void Domain::layersUpdateNeighCoords() {
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (!isOuterNeighInProc[pointI]) { outerNeighCoords[pointI] = UNDEF_VECTOR; continue; }
        const int neighI = pointToOuterPointMap[pointI];
        if (neighI < 0) { error = "Sanity broken, neighI does not exist for pointI"; return; }
        outerNeighCoords[pointI] = points[neighI];
    }
}

// serial set-up, SM.C:2215-2221
void Domain::setupLayers(const std::vector<Patch>& p, const LayerParams& lp) {
    layersBegin(p, lp);
    if (!doLayerTreatment) return;
    const int maxIter = lay.maxLayers + 1;                                   // SM.C:2217
    for (int iter = 0; iter < maxIter; ++iter) layersHopsSweep();            // OBB.C:83-131
    calculateBoundaryPointNormals();                                         // SM.C:2219
    for (int iter = 1; iter < maxIter + 1; ++iter) layersPropagateSweep(iter);   // OBB.C:274-366
    layersUndo();
}

void Domain::phaseA() {
    // SM.C:2262 reset frozen points
    isFrozenPoint.assign(nPoints, 0);
    updateGeometry();
    // SM.C:2266 "Recalculate point normals" (the reference does it whether or not a treatment is enabled;
    // the normals are only consumed by the layer treatment here)
    if (doLayerTreatment || doBoundarySmoothing) layersNormalsAccumulate();   // MultiDomain::syncLayers follows (OBB.C:184-198); finish in phaseB

    // SM.C:108-131 (internal points only unless doBoundarySmoothing, SM.C:116)
    cellSum.assign(nPoints, ZERO_VECTOR);
    cellCount.assign(nPoints, 0);
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if ((!doBoundarySmoothing) && (!isInternalPoint[pointI])) continue;
        const std::vector<int>& pCells = pointCells[pointI];
        cellCount[pointI] = int(pCells.size());
        for (int celli : pCells) cellSum[pointI] += cellCentres[celli];
    }

    // SM.C:325-387 local closest edge points
    closest1.assign(nPoints, ZERO_VECTOR);
    closest2.assign(nPoints, ZERO_VECTOR);
    closest3.assign(nPoints, ZERO_VECTOR);
    hasCommonCell.assign(nPoints, 0);
    std::vector<double> edgeLengths;
    std::vector<int> sLabels;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const Vec3 cCoords = points[pointI];
        const std::vector<int>& pp = pointPoints[pointI];
        const int n = int(pp.size());
        edgeLengths.assign(n, 0.0);
        for (int i = 0; i < n; ++i) edgeLengths[i] = getPointDistance(points[pp[i]], cCoords);
        // Foam::sortedOrder = stable sort of indices by value
        sLabels.resize(n);
        std::iota(sLabels.begin(), sLabels.end(), 0);
        std::stable_sort(sLabels.begin(), sLabels.end(),
                         [&](int a, int b) { return edgeLengths[a] < edgeLengths[b]; });
        const int cLabel1 = findAppropriateClosestPointLabel(pp, sLabels, pointI, isInternalPoint, 0);
        const int cLabel2 = findAppropriateClosestPointLabel(pp, sLabels, pointI, isInternalPoint, 1);
        const int cLabel3 = findAppropriateClosestPointLabel(pp, sLabels, pointI, isInternalPoint, 2);
        if (cLabel1 < 0 || cLabel2 < 0) {
            error = "Failed to find cLabel1/cLabel2 for pointI " + std::to_string(pointI);  // SM.C:354-362
            return;
        }
        closest1[pointI] = points[pp[cLabel1]] - cCoords;
        closest2[pointI] = points[pp[cLabel2]] - cCoords;
        if (cLabel3 < 0) closest3[pointI] = UNDEF_VECTOR;
        else closest3[pointI] = points[pp[cLabel3]] - cCoords;
        hasCommonCell[pointI] = (findIndex(pointNeighPoints[pp[cLabel1]], pp[cLabel2]) >= 0) ? 1 : 0;
    }
    // the rank-local part of updateNeighCoords (SM.C:2286): it only needs the current coordinates
    if (doLayerTreatment) layersUpdateNeighCoords();
    if (doBoundarySmoothing) { boundaryLocalPre(); if (!error.empty()) return; }   // SM.C:2310, BPS.C:866 local halves
}

// SM.C:1135-1231 (with calcFaceCenter :1103-1130, findCellFacePair :1042-1097,
// calcMinMaxFinalProjectedAngle :1003-1037)
void Domain::calcMinMaxFaceAngleForEdge(int edgeI, double& minFaceAngle, double& maxFaceAngle, int pointI1,
                                        const Vec3& coords1, int pointI2, const Vec3& coords2) const {
    const std::vector<int>& eFaces = edgeFaces[edgeI];
    const int nF = int(eFaces.size());
    const int e0I = edges[edgeI][0];
    Vec3 e0 = points[e0I];
    if ((pointI1 >= 0) && (e0I == pointI1)) e0 = coords1;
    else if ((pointI2 >= 0) && (e0I == pointI2)) e0 = coords2;
    const int e1I = edges[edgeI][1];
    Vec3 e1 = points[e1I];
    if ((pointI1 >= 0) && (e1I == pointI1)) e1 = coords1;
    else if ((pointI2 >= 0) && (e1I == pointI2)) e1 = coords2;

    const Vec3 cCoords = 0.5 * (e0 + e1);
    const Vec3 eVec = (e1 - e0) / mag(e1 - e0);

    std::vector<Vec3> pVecs(nF, UNDEF_VECTOR);
    for (int i = 0; i < nF; ++i) {
        const int faceI = eFaces[i];
        // calcFaceCenter SM.C:1103-1130
        Vec3 center{0, 0, 0};
        for (int pointI : faces[faceI]) {
            if ((pointI1 >= 0) && (pointI == pointI1)) center += coords1;
            else if ((pointI2 >= 0) && (pointI == pointI2)) center += coords2;
            else center += points[pointI];
        }
        center /= double(faces[faceI].size());
        const Vec3 fCoords = center;
        const Vec3 cf = cCoords - fCoords;
        const double dotProd = dot(cf, eVec);
        const Vec3 pCoords = fCoords + dotProd * eVec;
        const Vec3 cp = (pCoords - cCoords) / mag(pCoords - cCoords);
        pVecs[i] = cp;
    }

    double minAngle = 2.0 * M_PI;
    double maxAngle = 0.0;
    for (int cellI : edgeCells[edgeI]) {
        // findCellFacePair
        int face0I = -1, face1I = -1;
        for (int faceI : cellFaces[cellI]) {
            const int faceIsI = findIndex(eFaces, faceI);
            if (faceIsI >= 0) {
                if (face0I == -1) face0I = faceIsI;
                else if (face1I == -1) face1I = faceIsI;
                else { const_cast<Domain*>(this)->error = "more than two edge faces belong to same cell"; return; }
            }
        }
        if (face0I == -1 || face1I == -1 || face0I == face1I) {
            const_cast<Domain*>(this)->error = "didn't find face pairs for cell " + std::to_string(cellI);
            return;
        }
        const Vec3 cellCenter = cellCentres[cellI];  // mesh.C()[cellI], SM.C:1218
        const Vec3 cf = cCoords - cellCenter;
        const double dotProd = dot(cf, eVec);
        const Vec3 pCoords = cellCenter + dotProd * eVec;
        const Vec3 cp = (pCoords - cCoords) / mag(pCoords - cCoords);
        const double angle = calcEdgeCenterEdgeAngle(pVecs[face0I], cp, pVecs[face1I]);
        if (angle < minAngle) minAngle = angle;
        if (angle > maxAngle) maxAngle = angle;
    }
    minFaceAngle = minAngle;
    maxFaceAngle = maxAngle;
}

// SM.C:1276-1308
void Domain::calcMinMaxFaceAngleForPoint(int pointI1, const Vec3& coords1, int pointI2, const Vec3& coords2,
                                         double& minFaceAngle, double& maxFaceAngle) const {
    minFaceAngle = 2.0 * M_PI;
    maxFaceAngle = 0.0;
    for (int edgeI : pointEdges[pointI1]) {
        double minAngle, maxAngle;
        calcMinMaxFaceAngleForEdge(edgeI, minAngle, maxAngle, pointI1, coords1, pointI2, coords2);
        if (minFaceAngle > minAngle) minFaceAngle = minAngle;
        if (maxFaceAngle < maxAngle) maxFaceAngle = maxAngle;
    }
}

void Domain::phaseB() {
    const std::vector<Vec3>& mp = points;
    if (doLayerTreatment || doBoundarySmoothing) layersNormalsFinish();   // OBB.C:201-230, after the plusEq syncs

    // SM.C:150-163
    centroidalPoints = points;
    for (int pointI = 0; pointI < nPoints; ++pointI)
        if (cellCount[pointI]) centroidalPoints[pointI] = cellSum[pointI] / double(cellCount[pointI]);

    // SM.C:566-590 aspectRatioSmoothing
    newPoints = centroidalPoints;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const double blendFrac = calcARSmoothingRatio(closest1[pointI], closest2[pointI], closest3[pointI],
                                                      hasCommonCell[pointI], isInternalPoint[pointI]);
        if (blendFrac > 0.0) {
            const Vec3 aCoords = mp[pointI] + (closest1[pointI] + closest2[pointI]) / 2.0;
            const Vec3 newCoords = (1.0 - blendFrac) * centroidalPoints[pointI] + blendFrac * aCoords;
            newPoints[pointI] = newCoords;
        }
    }
    arPoints = newPoints;

    // SM.C:684-754 constrainMaxStepLength(doGlobalScaling = false)
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const Vec3 cCoords = mp[pointI];
        const Vec3 stepDir = newPoints[pointI] - cCoords;
        double globalScale;
        if (mag(stepDir) > prm.maxStepLength) globalScale = prm.maxStepLength / (mag(stepDir) * prm.relStepFrac);
        else globalScale = 1.0;
        const Vec3 nCoords = cCoords + (prm.relStepFrac * globalScale) * stepDir;
        newPoints[pointI] = nCoords;
    }

    // SM.C:2283-2305 optional boundary layer treatment (updateNeighCoords OBB.C:464-500: its local part ran at the
    // end of phaseA, MultiDomain::syncLayers did the minMagSqr sync)
    if (doLayerTreatment) {
        // blendWithOrthogonalPoints OBB.C:507-567, called with maxLayers + 1 (SM.C:2299)
        const double layerMaxBlendingFraction = lay.layerMaxBlendingFraction;
        const double minLayers = lay.minLayers;
        const double maxLayers = lay.maxLayers + 1;
        for (int pointI = 0; pointI < nPoints; ++pointI) {
            if (pointNormals[pointI] == ZERO_VECTOR) continue;
            if (!isInternalPoint[pointI]) continue;
            const int nHops = pointHopsToLayerBoundary[pointI];
            if (nHops < 1) continue;
            const Vec3 pointNormal = pointNormals[pointI];
            const Vec3 outerNeighCoord = outerNeighCoords[pointI];
            if (outerNeighCoord == UNDEF_VECTOR) { error = "Sanity broken, outerNeighCoord is zero for pointI"; return; }
            const int maxHops = int(std::min(double(nHops - 1), maxLayers));   // label maxHops = min(label, double)
            const double length = lay.layerEdgeLength * std::pow(lay.layerExpansionRatio, double(maxHops));
            const double slope = -layerMaxBlendingFraction / (maxLayers - minLayers);
            const double y0 = -slope * maxLayers;
            const double y = y0 + slope * nHops;
            const double blendFrac = std::max(0.0, std::min(y, layerMaxBlendingFraction));
            const Vec3 newPoint = newPoints[pointI];
            const Vec3 orthoPoint = outerNeighCoord + length * pointNormal;
            const Vec3 blendedPoint = blendFrac * orthoPoint + (1.0 - blendFrac) * newPoint;
            newPoints[pointI] = blendedPoint;
        }
        // SM.C:2304 constrainMaxStepLength once more, over all points
        for (int pointI = 0; pointI < nPoints; ++pointI) {
            const Vec3 cCoords = mp[pointI];
            const Vec3 stepDir = newPoints[pointI] - cCoords;
            double globalScale;
            if (mag(stepDir) > prm.maxStepLength) globalScale = prm.maxStepLength / (mag(stepDir) * prm.relStepFrac);
            else globalScale = 1.0;
            const Vec3 nCoords = cCoords + (prm.relStepFrac * globalScale) * stepDir;
            newPoints[pointI] = nCoords;
        }
    }

    // SM.C:2307-2357 optional boundary point smoothing
    if (doBoundarySmoothing) {
        projectBoundaryPoints();
        if (!error.empty()) return;
    }

    // SM.C:602-652 restrictEdgeShortening
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (isFrozenPoint[pointI]) continue;
        const Vec3 cCoords = mp[pointI];
        const Vec3 nCoords = newPoints[pointI];
        double shortestCurrentEdgeLength = GREAT;
        double shortestNewEdgeLength = GREAT;
        for (int neighI : pointPoints[pointI]) {
            const double testCurrentLength = getPointDistance(mp[neighI], cCoords);
            if (testCurrentLength < shortestCurrentEdgeLength) shortestCurrentEdgeLength = testCurrentLength;
            const double testNewLength = getPointDistance(mp[neighI], nCoords);
            if (testNewLength < shortestNewEdgeLength) shortestNewEdgeLength = testNewLength;
        }
        const double shortestLength = std::min(shortestNewEdgeLength, shortestCurrentEdgeLength);
        if (prm.totalMinFreeze && (shortestLength < prm.minEdgeLength)) isFrozenPoint[pointI] = 1;
        else if ((shortestNewEdgeLength < prm.minEdgeLength) && (shortestNewEdgeLength < shortestCurrentEdgeLength))
            isFrozenPoint[pointI] = 1;
    }
    frozenAfterEdgeLen = isFrozenPoint;

    // SM.C:900-930 restrictMinEdgeAngleDecrease (+ calc_min_edge_angles :837-894,
    // getNeighbourPoints :793-831)
    eaMinC.assign(nPoints, 0.0);
    eaMinN.assign(nPoints, 0.0);
    if (prm.edgeAngleConstraint) {
        for (int pointI = 0; pointI < nPoints; ++pointI) {
            if (isFrozenPoint[pointI]) continue;
            double minCAngle = DBL_MAX;
            double minNAngle = DBL_MAX;
            for (int faceI : pointFaces[pointI]) {
                const std::vector<int>& facePoints = faces[faceI];
                const int nP = int(facePoints.size());
                int neighPI1 = 0, neighPI2 = 0;
                for (int i = 0; i < nP; ++i) {
                    if (facePoints[i] == pointI) {
                        const int prevI = (i == 0) ? nP - 1 : i - 1;
                        const int nextI = (i == nP - 1) ? 0 : i + 1;
                        neighPI1 = facePoints[prevI];
                        neighPI2 = facePoints[nextI];
                        break;
                    }
                }
                const Vec3 cp0 = mp[pointI];
                const Vec3 cp1 = mp[neighPI1];
                const Vec3 cp2 = mp[neighPI2];
                const double cAngle = edgeEdgeAngle(cp0, cp1, cp2);
                const Vec3 np0 = newPoints[pointI];
                const double nAngle0 = edgeEdgeAngle(np0, cp1, cp2);
                const Vec3 np1 = newPoints[neighPI1];
                const Vec3 np2 = newPoints[neighPI2];
                const double nAngle1 = edgeEdgeAngle(np0, np1, np2);
                const double nAngle2 = edgeEdgeAngle(np0, cp1, np2);
                const double nAngle3 = edgeEdgeAngle(np0, np1, cp2);
                const double nAngle = std::min(std::min(std::min(nAngle0, nAngle1), nAngle2), nAngle3);
                if (cAngle < minCAngle) minCAngle = cAngle;
                if (nAngle < minNAngle) minNAngle = nAngle;
            }
            eaMinC[pointI] = minCAngle;
            eaMinN[pointI] = minNAngle;
            const double smallAngle = M_PI * prm.minAngle / 180.0;
            if (lessC(minNAngle, smallAngle, 0) && lessC(minNAngle, minCAngle, 0)) isFrozenPoint[pointI] = 1;
        }
    }
    frozenAfterEdgeAngle = isFrozenPoint;

    // SM.C:1320-1437 restrictFaceAngleDeterioration
    if (prm.faceAngleConstraint) {
        const int nEdges = int(edges.size());
        edgeMinAngle.assign(nEdges, GREAT);
        edgeMaxAngle.assign(nEdges, GREAT);
        // SM.C:1252-1270
        for (int edgeI = 0; edgeI < nEdges; ++edgeI) {
            double minAngle, maxAngle;
            calcMinMaxFaceAngleForEdge(edgeI, minAngle, maxAngle, -1, ZERO_VECTOR, -1, ZERO_VECTOR);
            if (!error.empty()) return;
            edgeMinAngle[edgeI] = minAngle;
            edgeMaxAngle[edgeI] = maxAngle;
        }
        // SM.C:938-975
        pointMinAngle.assign(nPoints, 2.0 * M_PI);
        pointMaxAngle.assign(nPoints, 0.0);
        for (int edgeI = 0; edgeI < nEdges; ++edgeI) {
            for (int k = 0; k < 2; ++k) {
                const int pointI = edges[edgeI][k];
                if (pointMinAngle[pointI] > edgeMinAngle[edgeI]) pointMinAngle[pointI] = edgeMinAngle[edgeI];
                if (pointMaxAngle[pointI] < edgeMaxAngle[edgeI]) pointMaxAngle[pointI] = edgeMaxAngle[edgeI];
            }
        }
        // SM.C:1347-1434 stack walk
        std::stack<int> pointStack;
        for (int pointI = 0; pointI < nPoints; ++pointI) pointStack.push(pointI);
        std::vector<char> rangeSeen(nPoints, 0);
        const double smallAngle = M_PI * prm.minAngle / 180.0;
        const double largeAngle = M_PI * prm.maxAngle / 180.0;
        while (!pointStack.empty()) {
            const int pointI = pointStack.top();
            pointStack.pop();
            const int rangeCls = rangeSeen[pointI] ? -1 : 1;      // (a point pushed again is tested again: the census' class counts points)
            rangeSeen[pointI] = 1;
            if (greaterC(pointMinAngle[pointI], smallAngle, rangeCls) && lessC(pointMaxAngle[pointI], largeAngle, rangeCls)) continue;
            const Vec3 cCoords = mp[pointI];
            Vec3 nCoords = newPoints[pointI];
            if (isFrozenPoint[pointI]) nCoords = cCoords;
            if (nCoords != cCoords) {
                double newMinFaceAngle, newMaxFaceAngle;
                calcMinMaxFaceAngleForPoint(pointI, nCoords, -1, nCoords, newMinFaceAngle, newMaxFaceAngle);
                if ((lessC(newMinFaceAngle, smallAngle, 2) && lessC(newMinFaceAngle, pointMinAngle[pointI], 2)) ||
                    (greaterC(newMaxFaceAngle, largeAngle, 2) && greaterC(newMaxFaceAngle, pointMaxAngle[pointI], 2))) {
                    nCoords = cCoords;
                    isFrozenPoint[pointI] = 1;
                }
            }
            for (int neighPointI : pointPoints[pointI]) {
                const Vec3 neighCoords = newPoints[neighPointI];
                if (isFrozenPoint[neighPointI]) continue;
                if (neighCoords == mp[neighPointI]) continue;
                double newMinFaceAngle, newMaxFaceAngle;
                calcMinMaxFaceAngleForPoint(pointI, nCoords, neighPointI, neighCoords, newMinFaceAngle,
                                            newMaxFaceAngle);
                if ((lessC(newMinFaceAngle, smallAngle, 2) && lessC(newMinFaceAngle, pointMinAngle[pointI], 2)) ||
                    (greaterC(newMaxFaceAngle, largeAngle, 2) && greaterC(newMaxFaceAngle, pointMaxAngle[pointI], 2))) {
                    isFrozenPoint[neighPointI] = 1;
                    pointStack.push(neighPointI);
                }
            }
        }
    }
    frozenAfterFaceAngle = isFrozenPoint;
}

void Domain::phaseC() {
    // SM.C:2384-2392
    nFrozenLocal = 0;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        if (isFrozenPoint[pointI] || ((!isInternalPoint[pointI]) && (!isSmoothingSurfacePoint[pointI]))) {
            newPoints[pointI] = points[pointI];
            ++nFrozenLocal;
        }
    }
    // SM.C:1546-1565
    double maxStep = 0.0;
    for (int pointI = 0; pointI < nPoints; ++pointI) {
        const double distance = mag(newPoints[pointI] - points[pointI]) / prm.maxStepLength;
        if (distance > maxStep) maxStep = distance;
    }
    residualLocal = maxStep;
}

void Domain::commit() { points = newPoints; }  // SM.C:2399 mesh.movePoints

// SM.C:2257-2437 (single rank: every syncPointList/returnReduce is the identity)
int Domain::iterate(int nIters, double relTol, double* residuals, int* nFrozen) {
    int done = 0;
    for (int i = 0; i < nIters; ++i) {
        phaseA();
        if (!error.empty()) return -1;
        phaseB();
        if (!error.empty()) return -1;
        phaseC();
        const double res = residualLocal;
        if (residuals) residuals[i] = res;
        if (nFrozen) nFrozen[i] = nFrozenLocal;
        commit();
        ++done;
        if (res < relTol) break;  // SM.C:2401-2405
    }
    return done;
}

// ---- multi-domain (MPI ranks of the reference, emulated in one process) ---------------
// syncTools::syncPointList model (OpenFOAM is not in the reference tree; restated from OpenFOAM's
// syncTools::syncPointList -> globalMeshData::syncPointData -> globalMeshData::syncData, the form of every version the
// reference builds against: OpenFOAM.org 12, OpenFOAM.com v2312-v2506, Allwmake:47):
//  * the sharers of a point form one group; the MASTER is the sharer with the lowest processor number
//    (globalPoints sorts a point's (processor, index) pairs, the first is the master), the slaves follow in
//    ascending processor order;
//  * syncData pulls the slaves' values to the master, folds  x = master;  for slaves ascending: cop(x, slave),
//    and hands every sharer THE SAME x;
//  * plusEqOp: the sum in ascending processor order; orEqOp / maxEqOp: order free;
//  * minMagSqrEqOp (ops.H: x = (magSqr(x) <= magSqr(y) ? x : y)) and maxMagSqrEqOp (>=): an exact tie keeps
//    the LOWER processor's value -- on every sharer.  This is what lets isCloserPoint's "same distance,
//    different coordinates" case (SM.C:242-244) fire: a rank can receive another rank's equal-length vector.
// syncVariant 1 (kept as an A/B switch; rounds 1-3 of this repository used it) is the processor-patch form of
// OpenFOAM < 2.0: every sharer folds the others' values onto ITS OWN (cop(mine, nbr)), so a tie keeps the own value
// and the sharers can end with different vectors.
static Vec3 foldMagSqrOf(const Vec3* v, int n, int self, bool takeMax, int syncVariant) {
    auto keepX = [&](const Vec3& x, const Vec3& y) { return takeMax ? (magSqr(x) >= magSqr(y)) : (magSqr(x) <= magSqr(y)); };
    if (syncVariant == 1) {
        Vec3 x = v[self];
        for (int k = 0; k < n; ++k) {
            if (k == self) continue;
            x = keepX(x, v[k]) ? x : v[k];
        }
        return x;
    }
    Vec3 x = v[0];                                        // the master's value
    for (int k = 1; k < n; ++k) x = keepX(x, v[k]) ? x : v[k];   // slaves in ascending processor order
    return x;
}

// The three sequential closest-point syncs of findClosestPoints (SM.C:391-469) for the n ranks
// sharing one point (ascending processor order); arrays hold every sharer's local values on entry and its synced
// values on exit.
void combineClosest(int n, Vec3* r1, Vec3* r2, Vec3* r3, unsigned char* hc, int syncVariant) {
    auto fold = [&](const std::vector<Vec3>& v, int self) { return foldMagSqrOf(v.data(), n, self, false, syncVariant); };
    {   // position 1, SM.C:397-419
        std::vector<Vec3> sent(r1, r1 + n);
        for (int j = 0; j < n; ++j) {
            const Vec3 sv = fold(sent, j);
            if (isCloserPoint(sv, r1[j])) { r3[j] = r2[j]; r2[j] = r1[j]; r1[j] = sv; hc[j] = 0; }
        }
    }
    {   // position 2, SM.C:424-445
        std::vector<Vec3> sent(r2, r2 + n);
        for (int j = 0; j < n; ++j) {
            const Vec3 sv = fold(sent, j);
            if (isCloserPoint(sv, r2[j])) { r3[j] = r2[j]; r2[j] = sv; hc[j] = 0; }
        }
    }
    {   // position 3, SM.C:450-469
        std::vector<Vec3> sent(r3, r3 + n);
        for (int j = 0; j < n; ++j) {
            const Vec3 sv = fold(sent, j);
            if (isCloserPoint(sv, r3[j])) { r3[j] = sv; }
        }
    }
}

void MultiDomain::syncA() {
    for (const SharedPoint& sp : shared) {
        const int n = int(sp.domain.size());
        // SM.C:134-148
        Vec3 s = ZERO_VECTOR;
        int cnt = 0;
        for (int j = 0; j < n; ++j) {
            s += dom[sp.domain[j]]->cellSum[sp.local[j]];
            cnt += dom[sp.domain[j]]->cellCount[sp.local[j]];
        }
        for (int j = 0; j < n; ++j) {
            dom[sp.domain[j]]->cellSum[sp.local[j]] = s;
            dom[sp.domain[j]]->cellCount[sp.local[j]] = cnt;
        }
        // SM.C:391-478: three sequential min-magnitude syncs + or-sync
        std::vector<Vec3> r1(n), r2(n), r3(n);
        std::vector<unsigned char> hc(n);
        for (int j = 0; j < n; ++j) {
            const Domain* d = dom[sp.domain[j]];
            r1[j] = d->closest1[sp.local[j]];
            r2[j] = d->closest2[sp.local[j]];
            r3[j] = d->closest3[sp.local[j]];
            hc[j] = d->hasCommonCell[sp.local[j]];
        }
        combineClosest(n, r1.data(), r2.data(), r3.data(), hc.data(), syncVariant);
        unsigned char any = 0;  // SM.C:472-478
        for (int j = 0; j < n; ++j) any |= hc[j];
        for (int j = 0; j < n; ++j) {
            Domain* d = dom[sp.domain[j]];
            d->closest1[sp.local[j]] = r1[j];
            d->closest2[sp.local[j]] = r2[j];
            d->closest3[sp.local[j]] = r3[j];
            d->hasCommonCell[sp.local[j]] = any;
        }
    }
}

void MultiDomain::syncFrozen() {  // SM.C:2374-2380
    for (const SharedPoint& sp : shared) {
        unsigned char any = 0;
        for (size_t j = 0; j < sp.domain.size(); ++j) any |= dom[sp.domain[j]]->isFrozenPoint[sp.local[j]];
        for (size_t j = 0; j < sp.domain.size(); ++j) dom[sp.domain[j]]->isFrozenPoint[sp.local[j]] = any;
    }
}

// syncPointList(maxMagSqrEqOp / minMagSqrEqOp) for one shared point (sharers in ascending processor order): the
// master fold by default, every sharer's own fold with syncVariant 1 (see above)
static void foldMagSqr(std::vector<Vec3>& v, bool takeMax, int syncVariant) {
    const int n = int(v.size());
    const std::vector<Vec3> sent(v);
    for (int self = 0; self < n; ++self) v[self] = foldMagSqrOf(sent.data(), n, self, takeMax, syncVariant);
}

// OBB.C:184-198 plusEq syncs of calculateBoundaryPointNormals (sums in ascending rank order, as syncA does for the
// cell sums) and OBB.C:490-496 minMagSqr sync of updateNeighCoords
void MultiDomain::syncLayers() {
    if (dom.empty() || !(dom[0]->doLayerTreatment || dom[0]->doBoundarySmoothing)) return;
    for (const SharedPoint& sp : shared) {
        const int n = int(sp.domain.size());
        Vec3 s = ZERO_VECTOR;
        int cnt = 0;
        for (int j = 0; j < n; ++j) {
            s += dom[sp.domain[j]]->pointNormals[sp.local[j]];
            cnt += dom[sp.domain[j]]->layerNFaces[sp.local[j]];
        }
        std::vector<Vec3> nc(n);
        for (int j = 0; j < n; ++j) {
            dom[sp.domain[j]]->pointNormals[sp.local[j]] = s;
            dom[sp.domain[j]]->layerNFaces[sp.local[j]] = cnt;
            nc[j] = dom[sp.domain[j]]->outerNeighCoords[sp.local[j]];
        }
        foldMagSqr(nc, false, syncVariant);
        for (int j = 0; j < n; ++j) dom[sp.domain[j]]->outerNeighCoords[sp.local[j]] = nc[j];
    }
}

// SM.C:2215-2221 with the syncPointList calls of OBB.C:124-130 (maxEq), :184-198 (plusEq), :359-365 (maxMagSqr)
void MultiDomain::setupLayers(const std::vector<std::vector<Patch>>& p, const LayerParams& lp) {
    for (size_t d = 0; d < dom.size(); ++d) dom[d]->layersBegin(p[d], lp);
    if (dom.empty() || !dom[0]->doLayerTreatment) return;
    const int maxIter = lp.maxLayers + 1;
    for (int iter = 0; iter < maxIter; ++iter) {
        for (Domain* d : dom) d->layersHopsSweep();
        for (const SharedPoint& sp : shared) {
            int m = -1;
            for (size_t j = 0; j < sp.domain.size(); ++j) m = std::max(m, dom[sp.domain[j]]->pointHopsToLayerBoundary[sp.local[j]]);
            for (size_t j = 0; j < sp.domain.size(); ++j) dom[sp.domain[j]]->pointHopsToLayerBoundary[sp.local[j]] = m;
        }
    }
    for (Domain* d : dom) d->layersNormalsAccumulate();
    for (const SharedPoint& sp : shared) {
        Vec3 s = ZERO_VECTOR;
        int cnt = 0;
        for (size_t j = 0; j < sp.domain.size(); ++j) {
            s += dom[sp.domain[j]]->pointNormals[sp.local[j]];
            cnt += dom[sp.domain[j]]->layerNFaces[sp.local[j]];
        }
        for (size_t j = 0; j < sp.domain.size(); ++j) {
            dom[sp.domain[j]]->pointNormals[sp.local[j]] = s;
            dom[sp.domain[j]]->layerNFaces[sp.local[j]] = cnt;
        }
    }
    for (Domain* d : dom) d->layersNormalsFinish();
    for (int iter = 1; iter < maxIter + 1; ++iter) {
        for (Domain* d : dom) d->layersPropagateSweep(iter);
        for (const SharedPoint& sp : shared) {
            std::vector<Vec3> v(sp.domain.size());
            for (size_t j = 0; j < sp.domain.size(); ++j) v[j] = dom[sp.domain[j]]->pointNormals[sp.local[j]];
            foldMagSqr(v, true, syncVariant);
            for (size_t j = 0; j < sp.domain.size(); ++j) dom[sp.domain[j]]->pointNormals[sp.local[j]] = v[j];
        }
    }
    for (Domain* d : dom) d->layersUndo();
}

// Boundary point smoothing under -parallel.  What is rank-local in the reference stays rank-local here (classification by
// the rank's own patches and neighbours, inner neighbour map, feature projections of the rank's own neighbour points -- a
// neighbour that is itself shared is projected, and counted, by every rank that holds it); what the reference reduces or
// synchronises is combined over the ranks in ascending rank order.
void MultiDomain::setupBoundary(const std::vector<std::vector<Patch>>& p, const LayerParams& lp, const BoundaryInput& in) {
    if (dom.empty()) return;
    // getMeshStats reductions SM.C:1528-1538
    double minLen = VGREAT, bb[6] = {VGREAT, -VGREAT, VGREAT, -VGREAT, VGREAT, -VGREAT};
    for (Domain* d : dom) {
        double m, b[6];
        d->boundaryStatsLocal(m, b);
        if (m < minLen) minLen = m;
        for (int k = 0; k < 6; k += 2) { if (b[k] < bb[k]) bb[k] = b[k]; if (b[k + 1] > bb[k + 1]) bb[k + 1] = b[k + 1]; }
    }
    const double perimeter = bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4];
    for (size_t d = 0; d < dom.size(); ++d) {
        dom[d]->boundaryBegin(p[d], lp, in, minLen, perimeter);
        if (!dom[d]->error.empty()) return;
    }
    auto syncMax = [&](std::vector<int> Domain::*field) {   // maxEqOp, OBB.C:124-130
        for (const SharedPoint& sp : shared) {
            int m = -1;
            for (size_t j = 0; j < sp.domain.size(); ++j) m = std::max(m, (dom[sp.domain[j]]->*field)[sp.local[j]]);
            for (size_t j = 0; j < sp.domain.size(); ++j) (dom[sp.domain[j]]->*field)[sp.local[j]] = m;
        }
    };
    const int maxIter = lp.maxLayers + 1;
    for (int iter = 0; iter < maxIter; ++iter) {
        for (Domain* d : dom) d->layersHopsSweep();
        syncMax(&Domain::pointHopsToLayerBoundary);
    }
    for (int iter = 0; iter < 2; ++iter) {
        for (Domain* d : dom) d->boundaryHopsSweep();
        syncMax(&Domain::pointHopsToSmoothingBoundary);
    }
    for (Domain* d : dom) d->layersNormalsAccumulate();
    for (const SharedPoint& sp : shared) {   // plusEq of normals and face counts, OBB.C:184-198
        Vec3 s = ZERO_VECTOR;
        int cnt = 0;
        for (size_t j = 0; j < sp.domain.size(); ++j) {
            s += dom[sp.domain[j]]->pointNormals[sp.local[j]];
            cnt += dom[sp.domain[j]]->layerNFaces[sp.local[j]];
        }
        for (size_t j = 0; j < sp.domain.size(); ++j) {
            dom[sp.domain[j]]->pointNormals[sp.local[j]] = s;
            dom[sp.domain[j]]->layerNFaces[sp.local[j]] = cnt;
        }
    }
    for (Domain* d : dom) d->layersNormalsFinish();
    for (int iter = 1; iter < maxIter + 1; ++iter) {
        for (Domain* d : dom) d->layersPropagateSweep(iter);
        for (const SharedPoint& sp : shared) {   // maxMagSqr, OBB.C:359-365
            std::vector<Vec3> v(sp.domain.size());
            for (size_t j = 0; j < sp.domain.size(); ++j) v[j] = dom[sp.domain[j]]->pointNormals[sp.local[j]];
            foldMagSqr(v, true, syncVariant);
            for (size_t j = 0; j < sp.domain.size(); ++j) dom[sp.domain[j]]->pointNormals[sp.local[j]] = v[j];
        }
    }
    for (Domain* d : dom) { d->layersUndo(); d->boundaryFinish(); }
}

void MultiDomain::syncBoundary() {
    if (dom.empty() || !dom[0]->doBoundarySmoothing) return;
    for (const SharedPoint& sp : shared) {
        const int n = int(sp.domain.size());
        Vec3 s = ZERO_VECTOR;
        int cnt = 0;
        std::vector<Vec3> nc(n);
        for (int j = 0; j < n; ++j) {
            s += dom[sp.domain[j]]->featureEdgeProjections[sp.local[j]];       // plusEqOp, BPS.C:659-674
            cnt += dom[sp.domain[j]]->nFeatureEdgeProjections[sp.local[j]];
            nc[j] = dom[sp.domain[j]]->innerNeighCoords[sp.local[j]];
        }
        foldMagSqr(nc, false, syncVariant);                                                 // minMagSqrEqOp, OBB.C:490-496
        for (int j = 0; j < n; ++j) {
            dom[sp.domain[j]]->featureEdgeProjections[sp.local[j]] = s;
            dom[sp.domain[j]]->nFeatureEdgeProjections[sp.local[j]] = cnt;
            dom[sp.domain[j]]->innerNeighCoords[sp.local[j]] = nc[j];
        }
    }
}

int MultiDomain::iterate(int nIters, double relTol, double* residuals, int* nFrozen) {
    int done = 0;
    for (int i = 0; i < nIters; ++i) {
        for (Domain* d : dom) { d->phaseA(); if (!d->error.empty()) return -1; }
        syncA();
        syncLayers();
        syncBoundary();
        for (Domain* d : dom) { d->phaseB(); if (!d->error.empty()) return -1; }
        syncFrozen();
        double res = 0.0;
        int nf = 0;
        for (Domain* d : dom) {
            d->phaseC();
            if (d->residualLocal > res) res = d->residualLocal;  // returnReduce max, SM.C:1567
            nf += d->nFrozenLocal;                               // returnReduce sum, SM.C:2396
        }
        if (residuals) residuals[i] = res;
        if (nFrozen) nFrozen[i] = nf;
        for (Domain* d : dom) d->commit();
        ++done;
        if (res < relTol) break;
    }
    return done;
}

}  // namespace orc
