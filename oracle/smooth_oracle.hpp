// smooth_oracle.hpp -- CPU restatement of tkeskita/smoothMesh's centroidal smoothing
// iteration loop.  TEST INFRASTRUCTURE ONLY: this is the parity oracle and the timed
// CPU baseline.  Nothing under smoothmesh_amd/ (the product) may include, link or
// call anything in this directory; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do.
//
// PARITY UNPINNED: the reference (OpenFOAM application, src/smoothMesh.C) cannot be
// compiled or run in this environment (no OpenFOAM, no MPI) and ships no golden
// vectors (run_tests.sh only checks for crashes).  This restatement follows the
// reference function by function (citations below and in the .cpp) and is pinned
// only by analytic known answers and invariances (tests/test_oracle_*.py).
//
// Reference citations are relative to /root/reference/ (SM.C = src/smoothMesh.C,
// BPS.C = src/boundaryPointSmoothing.C, COM.H = src/smoothMeshCommon.H).
// OpenFOAM's primitiveMesh geometry/addressing is third-party code that is NOT in the
// reference tree; it is restated from OpenFOAM (.com v2412 variant, one of the
// versions accepted by Allwmake:47): primitiveMeshFaceCentresAndAreas.C,
// primitiveMeshCellCentresAndVols.C, primitiveMeshEdges.C, primitiveMeshPointCells.C.
#pragma once
#include <array>
#include <string>
#include <vector>

namespace orc {

// acos variant (process-wide): 0 = std::acos (glibc: the reference's arithmetic, default), 1 = the device kernels' algorithm
// (smoothmesh_amd/csrc/smacos.hpp, bit-identical on CPU and GPU)
void setAcosVariant(int v);
int acosVariant();
// census of the threshold comparisons of angles (smooth_oracle.cpp)
// minUlp: over the unequal pairs.  near[c]: comparisons with sides 1 .. window ulp apart, by class as the engine's near-tie census
// counts them (include/smgpu.h, smgpu_get_near_ties): 0 = SM.C:923, 1 = SM.C:1367 (once per point and iteration), 2 = SM.C:1391-1394 / 1421-1424
struct AcosCensus { long long comparisons = 0, equal = 0, within8ulp = 0; unsigned long long minUlp = ~0ull; long long near[3] = {0, 0, 0}; };
void censusWindow(unsigned long long ulps);
void censusEnable(bool on);
void censusReset();
AcosCensus censusGet();


// OpenFOAM double-precision constants (doubleScalar.H), used at SM.C:259,621,1333,1486.
constexpr double GREAT = 1.0e+15;
constexpr double VGREAT = 1.0e+300;
constexpr double SMALL = 1.0e-15;
constexpr double VSMALL = 1.0e-300;
constexpr double ROOTVSMALL = 1.0e-150;

struct Vec3 {
    double x, y, z;
};

struct Params {
    double maxStepLength = 0.0;   // SM.C:1864
    double relStepFrac = 0.5;     // SM.C:1874
    double minEdgeLength = 0.0;   // SM.C:1861
    bool totalMinFreeze = false;  // SM.C:1877
    bool edgeAngleConstraint = true;  // SM.C:1886
    bool faceAngleConstraint = true;  // SM.C:1889
    double minAngle = 35.0;       // SM.C:1880 (degrees)
    double maxAngle = 160.0;      // SM.C:1883 (degrees)
};

// polyMesh boundary patch as the layer treatment needs it (OBB.C = src/orthogonalBoundaryBlending.C)
struct Patch {
    int start = 0, size = 0;   // face range
    int kind = 0;              // 0 ordinary, 1 processor, 2 empty (skipped by OBB.C:156-159)
    bool isLayerPatch = false; // selected by -layerPatches (SM.C:1823)
    bool isSmoothingPatch = false;   // selected by -smoothingPatches (SM.C:1837-1842; default: all patches)
};

// OpenFOAM edgeMesh as the boundary point smoothing uses it: points, edges, pointEdges (ascending edge ids)
struct EdgeMesh {
    std::vector<Vec3> points;
    std::vector<std::array<int, 2>> edges;
    std::vector<std::vector<int>> pointEdges;
    void buildPointEdges();
};
struct TriSurface {
    std::vector<Vec3> points;
    std::vector<std::array<int, 3>> tris;
};
// inputs of the optional boundary point smoothing: constant/geometry/{initEdges,targetEdges,targetSurfaces}.obj
// (SM.C:1924-1926) and the classification lists a previous run left (SM.C:2039-2077; empty = no data)
struct BoundaryInput {
    EdgeMesh initEdges, targetEdges;
    TriSurface surf;
    std::vector<int> isCornerPointIO, isFeatureEdgePointIO;
    double internalSmoothingBlendingFraction = 0.0;   // SM.C:1907
};

// boundary layer treatment options, defaults SM.C:1892-1905
struct LayerParams {
    double layerMaxBlendingFraction = 0.3;
    double layerEdgeLength = 0.0;      // default: minEdgeLength, SM.C:1895
    double layerExpansionRatio = 1.3;
    int minLayers = 1, maxLayers = 4;
};

// One mesh (= one MPI rank's sub-domain in the reference).
class Domain {
public:
    // topology + coordinates (polyMesh content)
    int nPoints = 0, nCells = 0, nFaces = 0, nInternalFaces = 0;
    std::vector<Vec3> points;
    std::vector<std::vector<int>> faces;  // face -> point loop
    std::vector<int> owner;               // size nFaces (polyMesh::faceOwner)
    std::vector<int> neighbour;           // size nInternalFaces
    std::vector<unsigned char> isInternalPoint;          // SM.C:40-91
    std::vector<unsigned char> isSmoothingSurfacePoint;  // BPS.C:404-412 (all false when
                                                         // boundary smoothing is off)
    Params prm;

    // derived addressing (OpenFOAM primitiveMesh orderings)
    std::vector<std::array<int, 2>> edges;
    std::vector<std::vector<int>> pointCells, pointFaces, pointEdges, pointPoints;
    std::vector<std::vector<int>> edgeFaces, edgeCells, cellFaces, cellPoints;
    std::vector<std::vector<int>> pointNeighPoints;  // SM.C:190-217

    // geometry (OpenFOAM primitiveMesh)
    std::vector<Vec3> faceCentres, faceAreas, cellCentres;

    // per-iteration fields kept for operator-level parity checks
    std::vector<Vec3> cellSum;            // SM.C:108 "cellPoints"
    std::vector<int> cellCount;           // SM.C:110 "nPoints"
    std::vector<Vec3> closest1, closest2, closest3;  // SM.C:559-561
    std::vector<unsigned char> hasCommonCell;        // SM.C:564
    std::vector<Vec3> centroidalPoints;   // SM.C:2269
    std::vector<Vec3> arPoints;           // after aspectRatioSmoothing, SM.C:2276
    std::vector<Vec3> newPoints;          // proposal after constrainMaxStepLength, SM.C:2280
    std::vector<unsigned char> isFrozenPoint;        // SM.C:2009
    std::vector<unsigned char> frozenAfterEdgeLen, frozenAfterEdgeAngle, frozenAfterFaceAngle;
    std::vector<double> edgeMinAngle, edgeMaxAngle;    // SM.C:1333-1335
    std::vector<double> pointMinAngle, pointMaxAngle;  // SM.C:1339-1341
    std::vector<double> eaMinC, eaMinN;                // SM.C:914-916 per point
    int nFrozenLocal = 0;
    double residualLocal = 0.0;
    std::string error;

    // optional boundary layer treatment (serial): setup SM.C:2186-2221, per iteration SM.C:2266, 2283-2305
    std::vector<Patch> patches;
    LayerParams lay;
    bool doLayerTreatment = false;     // SM.C:2024-2028
    std::vector<unsigned char> isConnectedToInternalPoint, isLayerSurfacePoint;   // BPS.C:332-340, 397-403
    std::vector<unsigned char> isSharpEdgePoint, isOuterNeighInProc;
    std::vector<int> pointHopsToLayerBoundary, pointToOuterPointMap;
    std::vector<Vec3> pointNormals, outerNeighCoords;
    void setupLayers(const std::vector<Patch>& p, const LayerParams& lp);
    void calculateBoundaryPointNormals();   // OBB.C:141-233 (uses the current faceAreas)
    // the same in steps, so that MultiDomain can put the reference's syncPointList calls between them
    std::vector<int> layerNewHopCounts, layerNFaces, layerFirstMapper;
    void layersBegin(const std::vector<Patch>& p, const LayerParams& lp);  // SM.C:1983-2028, BPS.C:296-403, OBB.C:62-79
    void layersHopsSweep();                 // OBB.C:85-121 (one sweep, before the maxEq sync :124-130)
    void layersNormalsAccumulate();         // OBB.C:150-181 (before the plusEq syncs :184-198)
    void layersNormalsFinish();             // OBB.C:201-230
    void layersPropagateSweep(int iter);    // OBB.C:276-353 (one sweep, before the maxMagSqr sync :359-365)
    void layersUndo();                      // OBB.C:370-379
    void layersUpdateNeighCoords();         // OBB.C:471-486 (before the minMagSqr sync :490-496)

    // optional boundary point smoothing (serial): set-up SM.C:2080-2253, per iteration SM.C:2266-2269, 2307-2357
    // (BPS.C = src/boundaryPointSmoothing.C, OBB.C = src/orthogonalBoundaryBlending.C).  The ray query of
    // OpenFOAM's indexedOctree (findLine) is third-party code absent from the reference tree: restated as "nearest
    // hit along the segment" with OpenFOAM's triangle::intersection (Moller-Trumbore, HALF_RAY, tolerance
    // indexedOctree::perturbTol = 10*SMALL), ties to the lowest triangle id.
    bool doBoundarySmoothing = false;
    BoundaryInput bnd;
    double distanceTolerance = 0.0, meshMinEdgeLength = 0.0, meshPerimeter = 0.0;
    std::vector<int> targetEdgeStrings, pointStrings;
    std::vector<unsigned char> isFeatureEdgePoint, isCornerPoint, isFrozenSurfacePoint, isInnerNeighInProc;
    std::vector<int> isCornerPointOut, isFeatureEdgePointOut;   // the labelIOLists written back (BPS.C:370-379)
    std::vector<Vec3> cornerPoints, innerNeighCoords;
    std::vector<int> pointHopsToSmoothingBoundary, pointToInnerPointMap;
    void setupBoundary(const std::vector<Patch>& p, const LayerParams& lp, const BoundaryInput& in);
    // the same in steps, so that MultiDomain can put the reference's reductions / syncPointList calls between them
    std::vector<int> smoothingNewHopCounts, nFeatureEdgeProjections;
    std::vector<Vec3> featureEdgeProjections;
    void boundaryStatsLocal(double& minLength, double bb[6]) const;   // getMeshStats SM.C:1478-1526 before the reductions
    void boundaryBegin(const std::vector<Patch>& p, const LayerParams& lp, const BoundaryInput& in, double minEdgeGlobal, double perimeterGlobal);
    void boundaryHopsSweep();     // OBB.C:85-121 for the smoothing patches (before the maxEq sync :124-130)
    void boundaryFinish();        // OBB.C:396-459, SM.C:2234-2249
    void boundaryLocalPre();      // per iteration, before the syncs: OBB.C:471-486 (inner maps), BPS.C:637-656
    Vec3 findLine(const Vec3& start, const Vec3& end, bool& hit) const;
    Vec3 findIntersection(const Vec3& origPoint, const Vec3& pointNormal, double searchDistance) const;   // BPS.C:682-745
    void projectBoundaryPoints();     // BPS.C:843-945 + OBB.C:573-631 + SM.C:2356

    void build();  // addressing from faces/owner/neighbour
    void meshStats(double& minEdge, double& maxEdge) const;  // SM.C:1478-1541

    void updateGeometry();           // OpenFOAM makeFaceCentresAndAreas + makeCellCentresAndVols
    // Which OpenFOAM's geometry: 0 = OpenFOAM.com v2312-v2506 (fan triangles weighted by their area magnitude), 1 = OpenFOAM.org 12
    // (fan triangles weighted by their area projected on the face normal; pyramid volumes clamped at vSmall).  The reference
    // builds against either (Allwmake:47, README.md:30); both formulas are restated from OpenFOAM sources that are not under
    // /root/reference.
    int foamVariant = 0;
    // rank-engine form of the shared-point combines (oracle_capi.cpp orc_halo_combine*): the syncPointList model, as MultiDomain::syncVariant
    int syncVariant = 0;
    void phaseA();                   // geometry, SM.C:108-131 partial sums, SM.C:325-387 local closest
    void phaseB();                   // SM.C:155-163, 580-590, 684-754, 602-652, 900-930, 1320-1437
    void phaseC();                   // SM.C:2384-2392 restore+count, SM.C:1556-1565 residual
    void commit();                   // mesh.movePoints, SM.C:2399

    // single-domain loop SM.C:2257-2437; returns iterations done
    int iterate(int nIters, double relTol, double* residuals, int* nFrozen);

    // building blocks (public so tests can call them on hand-made inputs)
    void calcMinMaxFaceAngleForEdge(int edgeI, double& minA, double& maxA, int pointI1,
                                    const Vec3& coords1, int pointI2, const Vec3& coords2) const;
    void calcMinMaxFaceAngleForPoint(int pointI1, const Vec3& coords1, int pointI2,
                                     const Vec3& coords2, double& minA, double& maxA) const;
};

// Several domains iterated in lock-step with OpenFOAM syncTools::syncPointList
// semantics at shared points (SM.C:134,142,402-478,2374) and returnReduce (SM.C:1567,2396).
struct SharedPoint {
    // one entry per global point that is shared by >= 2 domains
    std::vector<int> domain;  // ascending domain id
    std::vector<int> local;   // local point id in that domain
};

class MultiDomain {
public:
    std::vector<Domain*> dom;
    std::vector<SharedPoint> shared;
    // syncTools::syncPointList model: 0 = globalMeshData::syncData (master fold, every sharer receives the same value; OpenFOAM
    // >= 2.0, i.e. every version the reference builds against), 1 = every sharer folds onto its own value (see smooth_oracle.cpp)
    int syncVariant = 0;
    int iterate(int nIters, double relTol, double* residuals, int* nFrozen);
    void syncA();
    void syncFrozen();
    // boundary layer treatment under -parallel: the set-up with its syncs, and the two per-iteration syncs
    void setupLayers(const std::vector<std::vector<Patch>>& p, const LayerParams& lp);
    // boundary point smoothing under -parallel: the set-up with its reductions and syncs (SM.C:2080-2253), and the per-iteration
    // syncs of the inner neighbour coordinates (minMagSqr, OBB.C:490-496) and the feature edge projections (plusEq, BPS.C:659-674)
    void setupBoundary(const std::vector<std::vector<Patch>>& p, const LayerParams& lp, const BoundaryInput& in);
    void syncBoundary();
    void syncLayers();   // plusEq of normals / face counts (OBB.C:184-198), minMagSqr of neighbour coordinates (:490-496)
};

int findEdgeMeshStrings(std::vector<int>& targetEdgeStrings, const EdgeMesh& em);   // BPS.C:557-587 (pointEdges must be built)
double edgeEdgeAngle(const Vec3& c, const Vec3& p1, const Vec3& p2);   // SM.C:766-786
double calcEdgeCenterEdgeAngle(const Vec3& p0, const Vec3& cC, const Vec3& p1);  // SM.C:980-998
bool isCloserPoint(const Vec3& a, const Vec3& b);  // SM.C:246-272
void combineClosest(int n, Vec3* r1, Vec3* r2, Vec3* r3, unsigned char* hc, int syncVariant = 0);  // SM.C:391-469

}  // namespace orc
