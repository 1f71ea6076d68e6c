// tiles.hpp -- LDS staging tables.
//
// The gather kernels of the loop (cell centres from faces from points; per-point sums of cell
// centres and neighbour coordinates) read 24-byte elements through int32 index lists.  Done straight
// from global memory, every wave instruction touches dozens of cache lines and the kernels are bound
// by L1 line throughput, not HBM.  Instead the element range is cut into TILES (consecutive cells /
// points, one workgroup each); for every tile the host lists, once, the unique source elements the
// tile needs (ascending ids => coalesced block loads into LDS) and rewrites the index lists as
// 16-bit LDS-local indices.  In the kernels all random access then happens inside LDS and global
// memory sees only streaming reads.  Arithmetic and summation order are unchanged.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "topology.hpp"

namespace smgpu {

// ---- geometry: tile = consecutive cells; LDS holds the tile's points and faces --------------------
struct GeomTiles {
    int32_t nTiles = 0;
    std::vector<int32_t> cellBeg;   // nTiles+1
    std::vector<int32_t> tpOff;     // nTiles+1 -> tpIds
    std::vector<int32_t> tpIds;     // unique point ids per tile, ascending
    std::vector<int32_t> tfOff;     // nTiles+1 -> tile faces
    std::vector<int32_t> tfIds;     // global face id per tile face (ascending); bit 31 = this tile holds the owner cell
    std::vector<int32_t> tfpOff;    // (#tile faces + 1) -> tfpLoc
    std::vector<uint16_t> tfpLoc;   // LDS-local point index of every vertex of every tile face
    std::vector<uint16_t> cfLoc;    // per cellFacesGeom entry: LDS-local face index, bit 15 = neighbour side
    int32_t maxPoints = 0, maxFaces = 0, maxCells = 0;
    std::string build(const Topology& t, int32_t capCells, int32_t capPoints, int32_t capFaces);
};

// ---- smoothing: tile = consecutive points; LDS holds the cell centres and neighbour points -------
struct SmoothTiles {
    int32_t nTiles = 0;
    std::vector<int32_t> ptBeg;     // nTiles+1
    std::vector<int32_t> tcOff;     // nTiles+1 -> tcIds
    std::vector<int32_t> tcIds;     // unique cell ids per tile, ascending
    std::vector<int32_t> tnOff;     // nTiles+1 -> tnIds
    std::vector<int32_t> tnIds;     // unique point ids (the tile's points and their neighbours), ascending
    std::vector<uint16_t> pcLoc;    // per pointCells entry: LDS-local cell index
    std::vector<uint16_t> ppLoc;    // per pointPoints entry: LDS-local point index
    std::vector<uint16_t> selfLoc;  // per point: its own LDS-local point index
    // per pointPoints entry (p -> q_i): bit j set <=> neighbours i and j of p share a cell
    // (pointNeighPoints membership, SM.C:383); valid while valence <= 16, else the kernel intersects lists
    std::vector<uint16_t> pairShare;
    int32_t maxCells = 0, maxPoints = 0, maxTilePoints = 0;
    std::string build(const Topology& t, int32_t capTile, int32_t capCells, int32_t capPoints);
};

}  // namespace smgpu
