// topology.hpp -- flat CSR addressing derived from a polyMesh (faces / owner / neighbour).
//
// Replaces what the reference gets from OpenFOAM's demand-driven primitiveMesh addressing
// (mesh.pointCells(), pointPoints(), pointFaces(), pointEdges(), edges(), edgeFaces(),
// edgeCells(); call sites src/smoothMesh.C:121,328,623,850,953,1149,1207,1294,1406) plus the
// reference's own helper lists (generateCellFaces SM.C:1575-1620, generatePointNeighPoints
// SM.C:190-217 -- the latter is replaced by an on-the-fly pointCells intersection).
//
// Orderings follow OpenFOAM where results depend on them:
//   pointCells  ascending cell id          (sum order of SM.C:127-130)
//   edges       (min,max) pairs in upper-triangular order (calcEdges, unsorted-points branch)
//   pointEdges  ascending edge id, pointPoints[p][i] = other end of pointEdges[p][i]
//               (=> ascending neighbour id; stable-sort ties SM.C:345, walk order SM.C:1406)
//   pointFaces / edgeFaces ascending face id
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace smgpu {

struct Csr {
    std::vector<int32_t> off;  // n+1
    std::vector<int32_t> val;
    int64_t nnz() const { return (int64_t)val.size(); }
    int32_t rows() const { return (int32_t)off.size() - 1; }
};

struct Topology {
    int32_t nPoints = 0, nCells = 0, nFaces = 0, nInternalFaces = 0, nEdges = 0;
    Csr facePoints;            // copy of the input
    std::vector<int32_t> owner, neighbour;

    Csr pointCells;
    Csr pointFaces;            // val = face id
    std::vector<int32_t> pfPrev, pfNext;  // per pointFaces entry: previous / next vertex of the
                                          // point in that face (getNeighbourPoints SM.C:793-831)
    std::vector<uint8_t> pfPrevSlot, pfNextSlot;  // the same two vertices as positions in the point's
                                                  // pointPoints row (255 = not representable)
    std::vector<int32_t> edges;           // 2*nEdges (start < end)
    Csr pointEdges;            // val = edge id
    std::vector<int32_t> pointPoints;     // same offsets as pointEdges
    Csr edgeFaces;
    Csr edgeCells;
    std::vector<uint8_t> ecFace0, ecFace1;  // per edgeCells entry: the two edge faces (indices into the
                                            // edge's edgeFaces row) that belong to the cell
                                            // (findCellFacePair SM.C:1042-1097)
    // Ring order of the faces / cells around each edge (same offsets as edgeFaces / edgeCells): cell i
    // of the ring lies between ring faces i and i+1 (the last cell of a closed ring between the last and
    // the first face).  Only min/max over the cells consume the order (SM.C:1003-1037), so any order is
    // valid; the ring lets every face vector be formed once.  edgeRingOk[e] = 0 for edges whose cells do
    // not form one chain (non-manifold): kernels use the generic pair form there.
    std::vector<int32_t> ringFace, ringCell;
    std::vector<uint8_t> edgeRingOk;
    // cell -> faces in the accumulation order of OpenFOAM makeCellCentresAndVols: faces owned
    // (ascending), then faces neighboured (ascending); bit 31 set = cell is the face's neighbour
    Csr cellFacesGeom;
    int32_t maxFaceSize = 0, maxEdgeFaces = 0, maxPointCells = 0, maxPointPoints = 0;

    // returns empty string on success, else the error (reference: FatalError)
    // afterCells (optional) is called once facePoints, owner / neighbour and cellFacesGeom stand (they are not touched again):
    // the geometry tile tables only need those and can be built next to the rest of the addressing (smgpu_create);
    // afterPoints likewise once the point-based lists stand (pointFaces with prev / next, pointCells, edges, pointEdges /
    // pointPoints, maxPointPoints): all the smoothing tile tables read
    std::string build(int32_t nPoints, int32_t nCells, int32_t nFaces, int32_t nInternalFaces,
                      const int32_t* faceOffsets, const int32_t* facePts, const int32_t* owner,
                      const int32_t* neighbour, const std::function<void()>& afterCells = nullptr,
                      const std::function<void()>& afterPoints = nullptr);
};

// the device arrays of a device build that the kernels read as they are (MeshView): handed to the caller instead of being freed
struct DeviceTopologyArrays {
    struct Arr { void* p = nullptr; size_t bytes = 0; };
    Arr owner, neighbour;      // (not read by the loop's kernels: the device builds of the tile tables do, tiles_dev.hip)
    Arr faceOff, facePts, cfOff, cfVal, pcOff, pcVal, ppOff, ppPt, peEdge, pfOff, pfFace, pfPrev, pfNext, pfPrevSlot, pfNextSlot, ringFace, ringCell,
        edgeRingOk, edges, efOff, efFace, ecOff, ecCell, ecF0, ecF1;
    bool valid = false;
};

// The same addressing built on the device (topology_dev.hip: radix sorts of (row, value) keys + per-edge kernels) and copied into
// t.  0: done; 1: not handled there -- the caller runs Topology::build (meshes it reports an error for, edges with more than 16
// faces, points with more than 255 neighbours, lists beyond 2^30 entries -- decided BEFORE a hook is called); 2: a HIP error (why).
// The hooks are called while the lists still arrive on the host, each once the lists a tile-BOUNDARY pass reads stand there
// (tiles.hpp): afterCells -- facePoints, owner / neighbour, cellFacesGeom; afterEdges -- also edges, edgeFaces, edgeCells;
// afterPoints -- also pointCells, the pointEdges offsets, pointPoints.  The rest stands when the function returns: a host build
// of tile TABLES has to wait for that.  deferUnread (with keep): the rest stays on the device only -- nothing in the loop and
// nothing in the layer / boundary set-up reads it on the host once the tile tables were built on the device (1.9 + 1.5 GB for
// 10 M cells, 0.5 s of the set-up).  downloadDeferredLists fetches it into t for whoever asks after all: groups bit 0 --
// pointFaces with prev / next, pointEdges' edge ids (the shared points' tiles of a multi-rank run, the getters); bit 1 -- the
// prev / next slots, the edge cells' face pairs, the rings (checksums, a host build of the edge tables).
int buildTopologyOnDevice(Topology& t, int32_t nPoints, int32_t nCells, int32_t nFaces, int32_t nInternalFaces, const int32_t* faceOffsets,
                          const int32_t* facePts, const int32_t* owner, const int32_t* neighbour, int device, std::string& why,
                          const std::function<void()>& afterCells = nullptr, const std::function<void()>& afterPoints = nullptr,
                          DeviceTopologyArrays* keep = nullptr, const std::function<void()>& afterEdges = nullptr, bool deferUnread = false);
int downloadDeferredLists(Topology& t, const DeviceTopologyArrays& td, int device, std::string& why, int groups = 3);

}  // namespace smgpu
