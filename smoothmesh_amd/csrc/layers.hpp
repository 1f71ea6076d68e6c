// layers.hpp -- one-off host set-up of the optional boundary layer treatment (prismatic layers).
//
// Replaces the reference's preparation SM.C:2186-2221: the part of classifyBoundaryPoints that the treatment consumes
// (BPS.C:296-340, 397-403), calculatePointHopsToBoundary (OBB.C:52-133), calculateBoundaryPointNormals (OBB.C:141-233)
// and propagateOuterNeighInfo (OBB.C:244-391)  (OBB.C = src/orthogonalBoundaryBlending.C, BPS.C =
// src/boundaryPointSmoothing.C).  The per-iteration part (SM.C:2266, 2283-2305) runs inside the smoothing kernels
// (layerTreat, kernels.hpp).
//
// The set-up is a sequence of rank-local steps; under -parallel the reference synchronises the shared points between
// them (syncTools::syncPointList: maxEq after every hop sweep OBB.C:124-130, plusEq of normals and face counts
// OBB.C:184-198, maxMagSqr of normals after every propagation sweep OBB.C:359-365).  LayerBuilder exposes the steps so
// that the host can do exactly that; buildLayerSetup runs them back to back (serial run).
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

class LayerBuilder {
public:
    LayerSetup out;
    int maxIter = 0;        // maxLayers + 1 sweeps (SM.C:2217)
    // faceArea: OpenFOAM face area vectors (3 per face) of the coordinates the set-up is made for.
    std::string begin(const Topology& t, const uint8_t* isInternalPoint, const std::vector<LayerPatch>& patches, const double* faceArea,
                      double layerMaxBlendingFraction, double layerEdgeLength, double layerExpansionRatio, int minLayers, int maxLayers);
    void hopsSweep();                 // OBB.C:85-121
    void normalsAccumulate();         // OBB.C:150-181; fills nFaces
    void normalsFinish();             // OBB.C:201-230
    void propagateSweep(int iter);    // OBB.C:276-353
    void finish();                    // OBB.C:370-379 + the per-hop tables
    std::vector<int32_t> nFaces;      // boundary faces per point (synchronised with plusEq by the host)

private:
    const Topology* t_ = nullptr;
    const uint8_t* internal_ = nullptr;
    std::vector<LayerPatch> patches_;
    const double* faceArea_ = nullptr;
    double maxBlend_ = 0, edgeLength_ = 0, ratio_ = 0;
    int minLayers_ = 0, maxLayers_ = 0;
    std::vector<int32_t> fresh_, firstClaim_;
};

// all steps back to back (serial run)
std::string buildLayerSetup(const Topology& t, const uint8_t* isInternalPoint, const std::vector<LayerPatch>& patches,
                            const double* faceArea, double layerMaxBlendingFraction, double layerEdgeLength,
                            double layerExpansionRatio, int minLayers, int maxLayers, LayerSetup& out);

}  // namespace smgpu
