// boundary.hpp -- one-off host set-up of the optional boundary point smoothing.
//
// Replaces the reference's preparation SM.C:2080-2253 for this feature: sanity checks of the feature edge meshes
// (BPS.C:20-79), the edge strings of the target edge mesh (BPS.C:446-587), classifyBoundaryPoints with corner and
// feature edge detection (BPS.C:269-441), the hop counts to the smoothing patches (OBB.C:52-133 with maxIter 2), the
// inner neighbour map (OBB.C:396-459) and the target string of every feature edge point (SM.C:2234-2249)
// (BPS.C = src/boundaryPointSmoothing.C, OBB.C = src/orthogonalBoundaryBlending.C).  The per-iteration part
// (SM.C:2266, 2307-2357) runs in kernels_boundary.hpp.
//
// OpenFOAM's indexedOctree<treeDataTriSurface> (third-party, not in the reference tree) is replaced by a bounding
// volume hierarchy over the target triangles; the query semantics are stated in kernels_boundary.hpp (findLine).
// Under -parallel the reference synchronises hop counts, normals, feature projections and inner neighbour coordinates over
// the processor patches: the set-up below comes in steps so that the host can do that between them (smgpu_boundary_*).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "topology.hpp"

namespace smgpu {

struct BndPatch {
    int32_t start, size;   // face range of the patch
    int32_t kind;          // 0 ordinary, 1 processor, 2 empty
    bool isSmoothing;      // selected by -smoothingPatches (default: every patch, SM.C:1837-1840)
};

// OpenFOAM edgeMesh: points, edges, pointEdges (ascending edge ids per point)
struct EdgeMeshHost {
    std::vector<double> pts;        // 3 per point
    std::vector<int32_t> edges;     // 2 per edge
    std::vector<std::vector<int32_t>> pointEdges;
    int32_t nPoints() const { return (int32_t)(pts.size() / 3); }
    int32_t nEdges() const { return (int32_t)(edges.size() / 2); }
    void buildPointEdges();
};

struct BoundaryInputHost {
    EdgeMeshHost initEdges, targetEdges;   // targetEdges empty: the initial edges are the target (SM.C:2154-2160)
    std::vector<double> surfPts;           // 3 per point
    std::vector<int32_t> surfTris;         // 3 per triangle
    const int32_t* isCornerPointIO = nullptr;        // per mesh point, from a previous run (SM.C:2039-2077), or NULL
    const int32_t* isFeatureEdgePointIO = nullptr;
    double distanceTolerance = 0.0;        // SM.C:1921
    double meshMinEdgeLength = 0.0, meshPerimeter = 0.0;   // getMeshStats SM.C:1478-1541
};

// Bounding volume hierarchy over the target triangles.  build() makes a binary tree (median split on the longest
// centroid axis, <= 4 triangles per leaf) and collapses it into 8-wide nodes: one visit tests eight boxes that lie next
// to each other in memory, so a query costs a handful of dependent memory round trips instead of two per binary level.
struct Bvh {
    std::vector<double> box;       // binary tree, 6 per node: min xyz, max xyz (inflated: the traversal is conservative)
    std::vector<int32_t> link;     // binary tree, 2 per node: children (left, right), or (-(first + 1), count) for a leaf
    std::vector<float> wideBox;    // 48 per wide node, floats rounded outwards: lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]
    std::vector<int32_t> wideRef;  // 16 per wide node: ref[8] = child wide node, or first triangle (leaf order) of a leaf
                                   // child; cnt[8] = -1 unused slot, 0 child is a wide node, > 0 triangles of a leaf child
    std::vector<double> triVerts;  // 10 per triangle, leaf order: nine coordinates + the original id (bit pattern)
    std::vector<int32_t> triId;    // original triangle id, leaf order
    int32_t wideDepth = 0;
    void build(const std::vector<double>& pts, const std::vector<int32_t>& tris);
};

struct BoundarySetup {
    bool enabled = false;          // doBoundarySmoothing, SM.C:2080-2093
    std::vector<uint8_t> isConnectedToInternalPoint, isCornerPoint, isFeatureEdgePoint, isSmoothingSurfacePoint, isFrozenSurfacePoint;
    std::vector<int32_t> isCornerPointOut, isFeatureEdgePointOut;   // the labelIOLists to write back
    std::vector<double> cornerPoints;              // 3 per point (GREAT = none)
    std::vector<int32_t> pointStrings, hopsToSmoothingBoundary, innerMap, targetEdgeStrings;
    EdgeMeshHost target;                           // the resolved target edge mesh
    int32_t nCorner = 0, nFeature = 0, nSmoothingSurface = 0, nFrozenSurface = 0;
    std::vector<int32_t> hopsFresh;                // newHopCounts of the sweeps (OBB.C:82)
    double distanceTolerance = 0.0;
};

// findEdgeMeshStrings BPS.C:557-587 alone (pointEdges must be built); returns the number of strings
int32_t edgeMeshStrings(const EdgeMeshHost& em, std::vector<int32_t>& strings);

// points: the coordinates the set-up is made for (3 per mesh point).  Returns the reference's FatalError text, or "".
// The set-up in steps (under -parallel the host synchronises the hop counts of the shared points between the sweeps):
//   buildBoundarySetup    everything up to the zero hop counts on the smoothing patches (in.meshMinEdgeLength / meshPerimeter
//                         already reduced over the ranks)
//   boundarySetupHopsSweep x 2
//   boundarySetupFinish   inner neighbour map, target strings of the feature edge points
std::string buildBoundarySetup(const Topology& t, const uint8_t* isInternalPoint, const double* points,
                               const std::vector<BndPatch>& patches, const BoundaryInputHost& in, BoundarySetup& out);
void boundarySetupHopsSweep(const Topology& t, const uint8_t* isInternalPoint, BoundarySetup& out);
std::string boundarySetupFinish(const Topology& t, const double* points, BoundarySetup& out);
// all of it back to back (serial run)
std::string buildBoundarySetupSerial(const Topology& t, const uint8_t* isInternalPoint, const double* points,
                                     const std::vector<BndPatch>& patches, const BoundaryInputHost& in, BoundarySetup& out);

}  // namespace smgpu
