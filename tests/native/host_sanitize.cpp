// Sanitizer driver for the product's host-only code (no GPU): addressing build, tile tables, layer set-up on a
// polyhedral mesh.  Built and run by tests/test_host_sanitizers.py with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../smoothmesh_amd/csrc/host/polymesh_io.hpp"
#include "../../smoothmesh_amd/csrc/layers.hpp"
#include "../../smoothmesh_amd/csrc/tiles.hpp"
#include "../../smoothmesh_amd/csrc/topology.hpp"

namespace smhost { void genCavityMesh(int N, double radius, double shell, double jitter, uint64_t seed, PolyMeshData& out); }

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 14;
    smhost::PolyMeshData m;
    smhost::genCavityMesh(N, 0.25, 0.12, 0.15, 7, m);
    smgpu::Topology t;
    std::string err = t.build(m.nPoints(), m.nCells, m.nFaces(), m.nInternalFaces(), m.faceOffsets.data(), m.facePoints.data(),
                              m.owner.data(), m.neighbour.data());
    if (!err.empty()) { std::fprintf(stderr, "topology: %s\n", err.c_str()); return 1; }
    // internal points: not on a boundary face
    std::vector<uint8_t> internal((size_t)m.nPoints(), 1);
    for (int f = m.nInternalFaces(); f < m.nFaces(); ++f)
        for (int k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k) internal[(size_t)m.facePoints[k]] = 0;
    for (int morton = 0; morton < 2; ++morton) {
        smgpu::GeomTiles g;
        err = g.build(t, m.points.data(), morton != 0, 256, 128, 768, 512);
        if (!err.empty()) { std::fprintf(stderr, "geom tiles: %s\n", err.c_str()); return 1; }
        smgpu::SmoothTiles s;
        err = s.build(t, m.points.data(), internal.data(), morton != 0, 256, 512, 768);
        if (!err.empty()) { std::fprintf(stderr, "smooth tiles: %s\n", err.c_str()); return 1; }
        smgpu::EdgeTiles e;
        err = e.build(t, m.points.data(), morton != 0, 256, 768, 1280, 768);
        if (!err.empty()) { std::fprintf(stderr, "edge tiles: %s\n", err.c_str()); return 1; }
    }
    // layer set-up on the cavity wall with unit "areas" along x (the values do not matter for memory safety)
    std::vector<smgpu::LayerPatch> patches;
    for (const smhost::PatchInfo& p : m.patches) patches.push_back({p.startFace, p.nFaces, 0, p.name == "cavity"});
    std::vector<double> area(3 * (size_t)m.nFaces(), 0.0);
    for (int f = 0; f < m.nFaces(); ++f) area[3 * (size_t)f] = 1.0;
    smgpu::LayerSetup ls;
    err = smgpu::buildLayerSetup(t, internal.data(), patches, area.data(), 0.3, 0.01, 1.3, 1, 4, ls);
    if (!err.empty()) { std::fprintf(stderr, "layers: %s\n", err.c_str()); return 1; }
    long mapped = 0;
    for (int v : ls.outerMap) mapped += v >= 0;
    std::printf("ok points %d cells %d edges %d mapped %ld\n", m.nPoints(), m.nCells, t.nEdges, mapped);
    return 0;
}
