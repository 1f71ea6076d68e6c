// tiles.hpp -- LDS staging tables.
//
// The gather kernels of the loop (cell centres from faces from points; per-point sums of cell
// centres and neighbour coordinates) read 24-byte elements through int32 index lists.  Done straight
// from global memory, every wave instruction touches dozens of cache lines and every list walk is a
// chain of dependent loads: the kernels end up latency / L1-line bound, far from HBM.  Instead the
// element range is cut into TILES (consecutive cells / points, one workgroup each).  For every tile
// the host lists, once, the unique source elements the tile needs (ascending ids => coalesced block
// loads into LDS) and rewrites the index lists as 16-bit LDS-local indices in a sliced-ELL layout:
// fixed width per tile, 4 entries (8 bytes) per lane per load, lane-contiguous, so a thread fetches
// its whole adjacency with a couple of independent coalesced loads.  In the kernels all indexed
// access then happens inside LDS and global memory is only streamed.  Arithmetic and summation
// order are unchanged (entries keep the list order, 0xFFFF pads the tail).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "topology.hpp"

namespace smgpu {

constexpr uint16_t kEllPad = 0xFFFF;

// Z-curve (Morton) order of n elements given by 3 coordinates each: position -> element id, ties in id order
std::vector<int32_t> mortonOrderOf(int32_t n, const double* xyz);

// ---- geometry: tile = consecutive cells; LDS holds the tile's points and faces --------------------
struct GeomTiles {
    int32_t nTiles = 0, threads = 0;
    std::vector<int32_t> cellBeg;   // nTiles+1
    std::vector<int32_t> tpOff;     // nTiles+1 -> tpIds
    std::vector<int32_t> tpIds;     // unique point ids per tile, ascending
    std::vector<int32_t> tfOff;     // nTiles+1 -> tile faces
    std::vector<int32_t> tfIds;     // global face id per tile face (ascending); bit 31 = this tile holds the owner cell
    // face vertices: per tile face, `fw` (= multiple of 4, tile-uniform) LDS-local point indices, padded
    std::vector<int32_t> fvBase;    // nTiles -> offset into faceVerts (units of uint16)
    std::vector<uint8_t> fvWidth;   // nTiles: fw
    std::vector<uint16_t> faceVerts;
    // cell faces: sliced ELL, per tile width cw (multiple of 4); entry (j, t) at cfBase + ((j/4)*threads + t)*4 + j%4
    // value = LDS-local face index | 0x8000 when the cell is the face's neighbour
    std::vector<int32_t> cfBase;    // nTiles
    std::vector<uint8_t> cfWidth;   // nTiles
    std::vector<uint16_t> cellFaces;
    std::vector<uint8_t> tileFlags; // nTiles: bit0 every face of the tile is a quadrilateral, bit1 every cell has 6 faces
                                    // (wave-uniform selection of the unrolled kernel paths)
    int32_t maxPoints = 0, maxFaces = 0;
    // tile order: position -> cell id.  Natural order, or a Morton (Z-curve) order of the cell centres so
    // that a run of consecutive positions is a compact 3-D brick (fewer faces / points shared with other
    // tiles => less duplicated face work, smaller LDS footprint).  Results do not depend on it.
    std::vector<int32_t> order;
    // capWeighted: 3 x points + faceWeight x faces a tile may stage (its LDS footprint in doubles)
    std::string build(const Topology& t, const double* points, bool morton, int32_t threads, int32_t capCells,
                      int32_t capPoints, int32_t capFaces, int32_t capWeighted = 0x7fffffff, int32_t faceWeight = 7);
    // the two halves of build(): tile order + greedy boundaries (host), and the per-tile tables -- which tiles_dev.hip builds on
    // the device from the boundaries where it can (buildGeomTablesOnDevice)
    // cellOrder (optional): the cells' Morton order computed elsewhere (tiles_dev.hip: cellMortonOrderOnDevice, the same order)
    std::string buildBoundaries(const Topology& t, const double* points, bool morton, int32_t threads, int32_t capCells,
                                int32_t capPoints, int32_t capFaces, int32_t capWeighted = 0x7fffffff, int32_t faceWeight = 7,
                                const std::vector<int32_t>* cellOrder = nullptr);
    std::string buildTables(const Topology& t);
};
// the tables of a device build (tiles_dev.hip) that the kernels read where they were built; the caller owns the arrays
struct GeomTilesDev {
    struct Arr { void* p = nullptr; size_t bytes = 0; };
    Arr cellOrder, cellBeg, tpIds, tfIds, faceVerts, cellFaces, meta;
    bool valid = false;
};
struct DeviceTopologyArrays;
// 0: built (gt holds the offsets / widths / flags / tfIds / maxima the host reads, `out` the device arrays); 1: not handled there
// (the caller runs gt.buildTables); 2: a HIP error (why)
int cellMortonOrderOnDevice(const DeviceTopologyArrays& td, int32_t nCells, int32_t nPoints, const double* points, int device, std::vector<int32_t>& order, std::string& why);
int buildGeomTablesOnDevice(GeomTiles& gt, const DeviceTopologyArrays& td, int32_t nCells, int device, GeomTilesDev& out, std::string& why);

// ---- smoothing: tile = consecutive points; LDS holds the cell centres and neighbour points -------
struct SmoothTiles {
    int32_t nTiles = 0, threads = 0;
    std::vector<int32_t> ptBeg;     // nTiles+1
    std::vector<int32_t> tcOff;     // nTiles+1 -> tcIds
    std::vector<int32_t> tcIds;     // unique cell ids per tile, ascending
    std::vector<int32_t> tnOff;     // nTiles+1 -> tnIds
    std::vector<int32_t> tnIds;     // unique point ids (the tile's points and their neighbours), ascending
    std::vector<uint16_t> selfLoc;  // per tile position: the point's own LDS-local point index
    // sliced ELL (layout as GeomTiles::cellFaces)
    std::vector<int32_t> pcBase;    // nTiles
    std::vector<uint8_t> pcWidth;
    std::vector<uint16_t> pcEll;    // LDS-local cell index per pointCells entry
    std::vector<int32_t> ppBase;    // nTiles (shared by ppEll and pairEll)
    std::vector<uint8_t> ppWidth;
    std::vector<uint16_t> ppEll;    // LDS-local point index per pointPoints entry (bit 15 is set later for internal neighbours)
    // pairEll entry (p, i): bit j set <=> neighbours i and j of p share a cell (pointNeighPoints
    // membership, SM.C:383); valid while valence <= 16, else the kernel intersects pointCells lists
    std::vector<uint16_t> pairEll;
    // pfEll: per (point, face) the LDS-local indices of the previous and the next vertex of the point in the
    // face (getNeighbourPoints SM.C:793-831), two entries per face, width pfWidth (multiple of 4)
    std::vector<int32_t> pfBase;
    std::vector<uint8_t> pfWidth;
    std::vector<uint16_t> pfEll;
    int32_t maxCells = 0, maxPoints = 0;
    std::vector<int32_t> order;     // tile order: position -> point id (see GeomTiles::order)
    // isInternal (SM.C:40-91) marks bit 15 of the ppEll entries whose neighbour is an internal point (SM.C:294)
    // pointOrder (optional): mortonOrderOf(nPoints, points), computed by the caller (it also serves the edge tiles)
    // subset (optional): tiles over THESE points only, in the given order (multi-rank: the shared points get tiles of their own,
    // smgpu_halo_configure) -- `order`, `selfLoc` and the ELL rows then have subset->size() positions
    // capTotal: cells + points a tile may stage together (what its LDS footprint is made of; the kernels' occupancy is set by it)
    std::string build(const Topology& t, const double* points, const uint8_t* isInternal, bool morton, int32_t threads,
                      int32_t capCells, int32_t capPoints, const std::vector<int32_t>* pointOrder = nullptr,
                      const std::vector<int32_t>* subset = nullptr, int32_t capTotal = 0x7fffffff);
    // the two halves of build(), as GeomTiles'
    std::string buildBoundaries(const Topology& t, const double* points, bool morton, int32_t threads, int32_t capCells, int32_t capPoints,
                                const std::vector<int32_t>* pointOrder = nullptr, const std::vector<int32_t>* subset = nullptr, int32_t capTotal = 0x7fffffff);
    std::string buildTables(const Topology& t, const uint8_t* isInternal, bool subset = false);
};
struct SmoothTilesDev {
    struct Arr { void* p = nullptr; size_t bytes = 0; };
    Arr order, ptBeg, tcIds, tnIds, selfLoc, pcEll, ppEll, pairEll, pfEll, meta;
    bool valid = false;
};
// the corner chains of all points computed ahead of the tables (tiles_dev.hip: startCornerChains)
struct CornerChains { int* chPrev = nullptr; int* chNext = nullptr; int* bad = nullptr; int* big = nullptr; void* stream = nullptr; int32_t nPoints = 0; bool started = false; };
int startCornerChains(const DeviceTopologyArrays& td, int32_t nPoints, int device, CornerChains& c, std::string& why);
void releaseCornerChains(CornerChains& c);
// (maxPointPoints: Topology's; the neighbour-pair masks exist while it is <= 16; chains: taken over and freed, or NULL)
int buildSmoothTablesOnDevice(SmoothTiles& st, const DeviceTopologyArrays& td, int32_t nPoints, int32_t maxPointPoints, const uint8_t* isInternal, int device,
                              SmoothTilesDev& out, std::string& why, CornerChains* chains = nullptr);

// ---- face-angle filter: tile = edges (Morton order of the edge midpoints); LDS holds the points, the
// face vertex averages and the cell centres the tile's edges need ------------------------------------------
struct EdgeTiles {
    int32_t nTiles = 0, threads = 0;
    std::vector<int32_t> order;     // position -> edge id
    std::vector<int32_t> edgeBeg;   // nTiles+1
    std::vector<int32_t> tpOff, tpIds, tfOff, tfIds, tcOff, tcIds;   // unique points / faces / cells per tile
    std::vector<uint16_t> epLoc;    // 2 per tile position: LDS-local ids of the edge's end points
    std::vector<int32_t> efBase, ecBase;        // nTiles
    std::vector<uint8_t> efWidth, ecWidth;      // multiples of 4
    std::vector<uint16_t> efEll, ecEll;         // ring faces / ring cells (Topology::ringFace/ringCell), sliced ELL;
                                                // an edge without a ring (non-manifold) has an all-pad face row
    int32_t maxPoints = 0, maxFaces = 0, maxCells = 0;
    // pointOrder (optional, with morton): the edges follow the Z-curve of their START points (edges are stored grouped by start
    // point) instead of a sort of their own over the 3x as many edge midpoints -- a tile is still a compact cluster of edges
    std::string build(const Topology& t, const double* points, bool morton, int32_t threads, int32_t capPoints,
                      int32_t capFaces, int32_t capCells, const std::vector<int32_t>* pointOrder = nullptr, int32_t capTotal = 0x7fffffff);
    // the two halves of build(), as GeomTiles'
    std::string buildBoundaries(const Topology& t, const double* points, bool morton, int32_t threads, int32_t capPoints,
                                int32_t capFaces, int32_t capCells, const std::vector<int32_t>* pointOrder = nullptr, int32_t capTotal = 0x7fffffff);
    std::string buildTables(const Topology& t);
};
struct EdgeTilesDev {
    struct Arr { void* p = nullptr; size_t bytes = 0; };
    Arr order, edgeBeg, tpIds, tfIds, tcIds, epLoc, efEll, ecEll, meta;
    long long nTf = 0;
    bool valid = false;
};
int buildEdgeTablesOnDevice(EdgeTiles& et, const DeviceTopologyArrays& td, int32_t nEdges, int device, EdgeTilesDev& out, std::string& why);
int remapEdgeFaceIdsOnDevice(const int* geomTfIds, long long nGeomTf, int32_t nFaces, int* edgeTfIds, long long nEdgeTf, int device, std::string& why);

}  // namespace smgpu
