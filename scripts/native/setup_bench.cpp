// setup_bench.cpp -- host-only timing + checksums of smgpu_create's table builds (addressing, tile tables) on a mesh dump.
//   g++ -O2 -std=c++17 -pthread -I smoothmesh_amd/csrc scripts/native/setup_bench.cpp smoothmesh_amd/csrc/topology.cpp smoothmesh_amd/csrc/tiles.cpp -o scripts/native/setup_bench
//   scripts/native/setup_bench mesh.bin      (mesh.bin: scripts/dump_mesh.py)
// Prints one line per table with an FNV-1a checksum: two builds of the library must print identical lines.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <vector>

#include "tiles.hpp"
#include "topology.hpp"

using namespace smgpu;
static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <class T> static void sum(const char* name, const std::vector<T>& v) { std::printf("%-22s n=%zu h=%016llx\n", name, v.size(), (unsigned long long)fnv(v.data(), v.size() * sizeof(T))); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4) return 1;
    const int32_t nP = hdr[0], nC = hdr[1], nF = hdr[2], nIF = hdr[3];
    std::vector<double> pts(3 * (size_t)nP);
    std::vector<int32_t> fo((size_t)nF + 1), own((size_t)nF), nei((size_t)nIF);
    std::vector<uint8_t> internal((size_t)nP);
    size_t ok = std::fread(pts.data(), 8, pts.size(), f);
    ok += std::fread(fo.data(), 4, fo.size(), f);
    std::vector<int32_t> fp((size_t)fo[nF]);
    ok += std::fread(fp.data(), 4, fp.size(), f);
    ok += std::fread(own.data(), 4, own.size(), f);
    ok += std::fread(nei.data(), 4, nei.size(), f);
    ok += std::fread(internal.data(), 1, internal.size(), f);
    std::fclose(f);
    (void)ok;
    // As smgpu_create pipelines it (smgpu.hip): the geometry tile tables start from Topology::build's afterCells hook, the smoothing
    // tile tables from its afterPoints hook, both next to the rest of the addressing -- so that ThreadSanitizer and the checksum
    // comparison (tests/test_host_sanitizers.py) cover the overlapped path: a later edit that writes a member a hook's reader
    // still uses (maxFaceSize, pointPoints ...) shows up as a race or as a changed checksum.  SETUP_BENCH_PIPELINED=0: one after
    // the other (the A/B the checksums are compared with).
    const bool pipelined = !(std::getenv("SETUP_BENCH_PIPELINED") && std::atoi(std::getenv("SETUP_BENCH_PIPELINED")) == 0);
    Topology t;
    GeomTiles gt; SmoothTiles st; EdgeTiles et;
    std::future<std::string> fG, fS;
    double t0 = now();
    const std::string e = pipelined
        ? t.build(nP, nC, nF, nIF, fo.data(), fp.data(), own.data(), nei.data(),
                  [&] { fG = std::async(std::launch::async, [&] { return gt.build(t, pts.data(), true, 256, 128, 768, 512); }); },
                  [&] { fS = std::async(std::launch::async, [&] { return st.build(t, pts.data(), internal.data(), true, 256, 512, 768); }); })
        : t.build(nP, nC, nF, nIF, fo.data(), fp.data(), own.data(), nei.data());
    double t1 = now();
    if (!e.empty()) { if (fG.valid()) fG.wait(); if (fS.valid()) fS.wait(); std::printf("error %s\n", e.c_str()); return 1; }
    std::fprintf(stderr, "addressing %.2f s%s\n", t1 - t0, pipelined ? " (tile tables started from its hooks)" : "");
    if (!pipelined) {
        fG = std::async(std::launch::async, [&] { return gt.build(t, pts.data(), true, 256, 128, 768, 512); });
        fS = std::async(std::launch::async, [&] { return st.build(t, pts.data(), internal.data(), true, 256, 512, 768); });
    }
    auto fE = std::async(std::launch::async, [&] { return et.build(t, pts.data(), true, 256, 512, 768, 512); });
    const std::string e2 = fS.get();
    const std::string e1 = fG.get(), e3 = fE.get();
    double t2 = now();
    std::fprintf(stderr, "tiles %.2f s (%s|%s|%s)\n", t2 - t1, e1.c_str(), e2.c_str(), e3.c_str());
    sum("cellFacesGeom.val", t.cellFacesGeom.val); sum("pointFaces.val", t.pointFaces.val); sum("pfPrev", t.pfPrev); sum("pfNext", t.pfNext);
    sum("pointCells.off", t.pointCells.off); sum("pointCells.val", t.pointCells.val); sum("edges", t.edges); sum("pointEdges.val", t.pointEdges.val);
    sum("pointPoints", t.pointPoints); sum("pfPrevSlot", t.pfPrevSlot); sum("pfNextSlot", t.pfNextSlot); sum("edgeFaces.off", t.edgeFaces.off);
    sum("edgeFaces.val", t.edgeFaces.val); sum("edgeCells.off", t.edgeCells.off); sum("edgeCells.val", t.edgeCells.val); sum("ecFace0", t.ecFace0);
    sum("ecFace1", t.ecFace1); sum("ringFace", t.ringFace); sum("ringCell", t.ringCell); sum("edgeRingOk", t.edgeRingOk);
    std::printf("max %d %d %d %d %d\n", t.maxFaceSize, t.maxPointCells, t.maxPointPoints, t.maxEdgeFaces, t.nEdges);
    sum("g.order", gt.order); sum("g.cellBeg", gt.cellBeg); sum("g.tpOff", gt.tpOff); sum("g.tpIds", gt.tpIds); sum("g.tfOff", gt.tfOff); sum("g.tfIds", gt.tfIds);
    sum("g.fvBase", gt.fvBase); sum("g.fvWidth", gt.fvWidth); sum("g.faceVerts", gt.faceVerts); sum("g.cfBase", gt.cfBase); sum("g.cfWidth", gt.cfWidth);
    sum("g.cellFaces", gt.cellFaces); sum("g.tileFlags", gt.tileFlags); std::printf("g.max %d %d %d\n", gt.maxPoints, gt.maxFaces, gt.nTiles);
    sum("s.order", st.order); sum("s.ptBeg", st.ptBeg); sum("s.tcOff", st.tcOff); sum("s.tcIds", st.tcIds); sum("s.tnOff", st.tnOff); sum("s.tnIds", st.tnIds);
    sum("s.selfLoc", st.selfLoc); sum("s.pcBase", st.pcBase); sum("s.pcWidth", st.pcWidth); sum("s.pcEll", st.pcEll); sum("s.ppBase", st.ppBase);
    sum("s.ppWidth", st.ppWidth); sum("s.ppEll", st.ppEll); sum("s.pairEll", st.pairEll); sum("s.pfBase", st.pfBase); sum("s.pfWidth", st.pfWidth);
    sum("s.pfEll", st.pfEll); std::printf("s.max %d %d %d\n", st.maxCells, st.maxPoints, st.nTiles);
    sum("e.order", et.order); sum("e.edgeBeg", et.edgeBeg); sum("e.tpOff", et.tpOff); sum("e.tpIds", et.tpIds); sum("e.tfOff", et.tfOff); sum("e.tfIds", et.tfIds);
    sum("e.tcOff", et.tcOff); sum("e.tcIds", et.tcIds); sum("e.epLoc", et.epLoc); sum("e.efBase", et.efBase); sum("e.ecBase", et.ecBase);
    sum("e.efWidth", et.efWidth); sum("e.ecWidth", et.ecWidth); sum("e.efEll", et.efEll); sum("e.ecEll", et.ecEll);
    std::printf("e.max %d %d %d %d\n", et.maxPoints, et.maxFaces, et.maxCells, et.nTiles);
    return 0;
}
