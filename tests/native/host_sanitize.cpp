// Sanitizer driver for the product's host-only code (no GPU): addressing build, tile tables, layer set-up on a
// polyhedral mesh.  Built and run by tests/test_host_sanitizers.py with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../smoothmesh_amd/csrc/host/polymesh_io.hpp"
#include "../../smoothmesh_amd/csrc/layers.hpp"
#include "../../smoothmesh_amd/csrc/boundary.hpp"
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
    // boundary point smoothing set-up: the unit cube's twelve edges (4 segments each, built by hand), a two-triangle
    // "surface" per cube side, every patch smoothed; then the hierarchy over a larger strip of triangles
    {
        smgpu::BoundaryInputHost in;
        auto addPoint = [&](double x, double y, double z) { in.initEdges.pts.insert(in.initEdges.pts.end(), {x, y, z}); return in.initEdges.nPoints() - 1; };
        for (int axis = 0; axis < 3; ++axis)
            for (int u = 0; u < 2; ++u)
                for (int v = 0; v < 2; ++v) {
                    int prev = -1;
                    for (int a = 0; a <= 4; ++a) {
                        double c[3];
                        c[axis] = a / 4.0; c[(axis + 1) % 3] = u; c[(axis + 2) % 3] = v;
                        const int id = addPoint(c[0], c[1], c[2]);   // corners duplicated: three open strings meet nowhere -- fine here
                        if (prev >= 0) { in.initEdges.edges.push_back(prev); in.initEdges.edges.push_back(id); }
                        prev = id;
                    }
                }
        const double q[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
        for (auto& c : q) in.surfPts.insert(in.surfPts.end(), {c[0], c[1], c[2]});
        const int quads[6][4] = {{0, 3, 2, 1}, {4, 5, 6, 7}, {0, 1, 5, 4}, {2, 3, 7, 6}, {1, 2, 6, 5}, {0, 4, 7, 3}};
        for (auto& f : quads) in.surfTris.insert(in.surfTris.end(), {f[0], f[1], f[2], f[0], f[2], f[3]});
        in.distanceTolerance = 1e-6;
        in.meshMinEdgeLength = 0.01;
        in.meshPerimeter = 3.0;
        std::vector<smgpu::BndPatch> bp;
        for (const smhost::PatchInfo& p : m.patches) bp.push_back({p.startFace, p.nFaces, 0, true});
        smgpu::BoundarySetup bs;
        err = smgpu::buildBoundarySetupSerial(t, internal.data(), m.points.data(), bp, in, bs);
        if (!err.empty()) { std::fprintf(stderr, "boundary: %s\n", err.c_str()); return 1; }
        if (!bs.enabled || bs.nSmoothingSurface == 0) { std::fprintf(stderr, "boundary: not enabled\n"); return 1; }
        std::vector<double> sp;
        std::vector<int32_t> st;
        for (int i = 0; i < 300; ++i) {   // a strip, with repeated (coincident-centroid) triangles at the end
            const double x = (i < 280) ? i * 0.01 : 1.0;
            sp.insert(sp.end(), {x, 0, 0, x + 0.01, 0, 0, x, 1, 0.001 * (i % 7)});
            st.insert(st.end(), {3 * i, 3 * i + 1, 3 * i + 2});
        }
        smgpu::Bvh bvh;
        bvh.build(sp, st);
        if (bvh.triId.size() != 300 || bvh.wideRef.empty() || bvh.wideDepth < 1) { std::fprintf(stderr, "bvh: bad build\n"); return 1; }
        long leafTris = 0;
        for (size_t w = 0; w < bvh.wideRef.size() / 16; ++w)
            for (int c = 0; c < 8; ++c) {
                const int n = bvh.wideRef[16 * w + 8 + (size_t)c];
                if (n > 4) { std::fprintf(stderr, "bvh: leaf of %d triangles\n", n); return 1; }
                if (n > 0) leafTris += n;
            }
        if (leafTris != 300) { std::fprintf(stderr, "bvh: %ld triangles in leaves\n", leafTris); return 1; }
    }
    // polyMesh I/O (argv[2] = scratch directory): ascii and binary round trips, then the ascii lists cut short at many places --
    // every read of a damaged file must end in an exception, nothing else (the test runs this with several I/O threads and a
    // small grain, so the pieces of the several-thread readers / writers are a few records each)
    long refused = 0;
    if (argc > 2) {
        const std::string root = argv[2];
        for (int binary = 0; binary < 2; ++binary) {
            const std::string d = root + (binary ? "/bin/constant/polyMesh" : "/asc/constant/polyMesh");
            smhost::writePolyMesh(d, "constant", m, binary != 0, 17);
            smhost::PolyMeshData r;
            smhost::readPolyMesh(d, "", r);
            if (r.points != m.points || r.faceOffsets != m.faceOffsets || r.facePoints != m.facePoints || r.owner != m.owner ||
                r.neighbour != m.neighbour || r.nCells != m.nCells) { std::fprintf(stderr, "io: round trip differs (binary %d)\n", binary); return 1; }
        }
        const std::string d = root + "/asc/constant/polyMesh";
        for (const char* name : {"points", "faces", "owner", "neighbour"}) {
            const std::string file = d + "/" + name;
            std::string all;
            { FILE* f = std::fopen(file.c_str(), "rb"); if (!f) return 1; char buf[65536]; size_t k; while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, k); std::fclose(f); }
            const size_t body = all.find("\n(\n");
            for (int cutAt = 1; cutAt <= 12; ++cutAt) {
                const size_t len = body + (all.size() - body) * (size_t)cutAt / 13;
                { FILE* f = std::fopen(file.c_str(), "wb"); std::fwrite(all.data(), 1, len, f); std::fclose(f); }
                smhost::PolyMeshData r;
                try { smhost::readPolyMesh(d, "", r); std::fprintf(stderr, "io: %s cut at %zu was read\n", name, len); return 1; }
                catch (const std::exception&) { ++refused; }
            }
            { FILE* f = std::fopen(file.c_str(), "wb"); std::fwrite(all.data(), 1, all.size(), f); std::fclose(f); }
        }
        smhost::PolyMeshData r;
        smhost::readPolyMesh(d, "", r);
        if (r.points != m.points) { std::fprintf(stderr, "io: restored case differs\n"); return 1; }
    }
    std::printf("ok points %d cells %d edges %d mapped %ld refused %ld\n", m.nPoints(), m.nCells, t.nEdges, mapped, refused);
    return 0;
}
