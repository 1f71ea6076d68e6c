// layers.hpp -- one-off host set-up of the optional boundary layer treatment (prismatic layers).
//
// Replaces, for a serial run, the reference's preparation SM.C:2186-2221: the part of classifyBoundaryPoints that
// the treatment consumes (BPS.C:296-340, 397-403), calculatePointHopsToBoundary (OBB.C:52-133),
// calculateBoundaryPointNormals (OBB.C:141-233) and propagateOuterNeighInfo (OBB.C:244-391)
// (OBB.C = src/orthogonalBoundaryBlending.C, BPS.C = src/boundaryPointSmoothing.C).  The per-iteration part
// (SM.C:2266, 2283-2305) runs inside the smoothing kernels (layerTreat, kernels.hpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "topology.hpp"

namespace smgpu {

struct LayerPatch {
    int32_t start, size;   // face range of the patch
    int32_t kind;          // 0 ordinary, 1 processor, 2 empty
    bool isLayer;          // selected by -layerPatches
};

struct LayerSetup {
    std::vector<int32_t> hops;       // pointHopsToLayerBoundary, -1 = none
    std::vector<int32_t> outerMap;   // pointToOuterPointMap, -1 = none
    std::vector<double> normals;     // 3 per point; zero = no treatment
    std::vector<uint8_t> isConnectedToInternalPoint, isLayerSurfacePoint;
    std::vector<double> lengthOfHops, blendOfHops;   // per hop count 0..maxLayers+1 (OBB.C:545-553)
};

// faceArea: OpenFOAM face area vectors (3 per face) of the coordinates the set-up is made for.
std::string buildLayerSetup(const Topology& t, const uint8_t* isInternalPoint, const std::vector<LayerPatch>& patches,
                            const double* faceArea, double layerMaxBlendingFraction, double layerEdgeLength,
                            double layerExpansionRatio, int minLayers, int maxLayers, LayerSetup& out);

}  // namespace smgpu
