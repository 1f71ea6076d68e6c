// smhost_capi.cpp -- C-ABI (include/smhost.h) over polymesh_io.
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../../include/smhost.h"
#include "polymesh_io.hpp"

using namespace smhost;
namespace smhost {
void genCavityMesh(int N, double radius, double shell, double jitter, uint64_t seed, PolyMeshData& out);
void genCavitySubdomain(int N, double radius, double shell, double jitter, uint64_t seed, const int grid[3], int rank, PolyMeshData& out,
                        std::vector<int64_t>* pointGlobal, std::vector<int64_t>* cellGlobal);
}

static thread_local std::string g_err;
struct smhost_mesh { PolyMeshData d; std::vector<int64_t> pointGlobal, cellGlobal; };

template <typename F>
static int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
}

extern "C" {
const char* smhost_last_error(void) { return g_err.c_str(); }

int smhost_read_polymesh(const char* dir, const char* pointsDir, smhost_mesh** out) {
    return guarded([&] {
        auto* m = new smhost_mesh();
        try { readPolyMesh(dir, pointsDir ? pointsDir : "", m->d); }
        catch (...) { delete m; throw; }
        *out = m;
    });
}
void smhost_mesh_free(smhost_mesh* m) { delete m; }

int smhost_mesh_sizes(const smhost_mesh* m, int32_t* nPoints, int32_t* nCells, int32_t* nFaces, int32_t* nInternalFaces,
                      int32_t* nPatches, int64_t* nnz) {
    *nPoints = m->d.nPoints(); *nCells = m->d.nCells; *nFaces = m->d.nFaces(); *nInternalFaces = m->d.nInternalFaces();
    *nPatches = (int32_t)m->d.patches.size(); *nnz = (int64_t)m->d.facePoints.size();
    return 0;
}
int smhost_mesh_copy(const smhost_mesh* m, double* points, int32_t* faceOffsets, int32_t* facePoints, int32_t* owner, int32_t* neighbour) {
    const auto& d = m->d;
    std::memcpy(points, d.points.data(), d.points.size() * sizeof(double));
    std::memcpy(faceOffsets, d.faceOffsets.data(), d.faceOffsets.size() * sizeof(int32_t));
    std::memcpy(facePoints, d.facePoints.data(), d.facePoints.size() * sizeof(int32_t));
    std::memcpy(owner, d.owner.data(), d.owner.size() * sizeof(int32_t));
    std::memcpy(neighbour, d.neighbour.data(), d.neighbour.size() * sizeof(int32_t));
    return 0;
}
int smhost_mesh_patch(const smhost_mesh* m, int32_t i, char* name, int32_t nameCap, char* type, int32_t typeCap, int32_t* nFaces,
                      int32_t* startFace, int32_t* myProcNo, int32_t* neighbProcNo) {
    if (i < 0 || i >= (int32_t)m->d.patches.size()) { g_err = "patch index out of range"; return 1; }
    const auto& p = m->d.patches[i];
    std::snprintf(name, nameCap, "%s", p.name.c_str());
    std::snprintf(type, typeCap, "%s", p.type.c_str());
    *nFaces = p.nFaces; *startFace = p.startFace; *myProcNo = p.myProcNo; *neighbProcNo = p.neighbProcNo;
    return 0;
}

int smhost_write_polymesh(const char* dir, const char* location, int32_t nPoints, const double* points, int32_t nFaces,
                          const int32_t* faceOffsets, const int32_t* facePoints, const int32_t* owner, int32_t nInternalFaces,
                          const int32_t* neighbour, int32_t nCells, int32_t nPatches, const char* const* names,
                          const char* const* types, const int32_t* pNFaces, const int32_t* pStart, const int32_t* pMy,
                          const int32_t* pNbr, int32_t binary, int32_t precision) {
    return guarded([&] {
        PolyMeshData d;
        d.points.assign(points, points + 3 * (size_t)nPoints);
        d.faceOffsets.assign(faceOffsets, faceOffsets + nFaces + 1);
        d.facePoints.assign(facePoints, facePoints + faceOffsets[nFaces]);
        d.owner.assign(owner, owner + nFaces);
        d.neighbour.assign(neighbour, neighbour + nInternalFaces);
        d.nCells = nCells;
        for (int32_t i = 0; i < nPatches; ++i) {
            PatchInfo p;
            p.name = names[i]; p.type = types[i]; p.nFaces = pNFaces[i]; p.startFace = pStart[i];
            p.myProcNo = pMy ? pMy[i] : -1; p.neighbProcNo = pNbr ? pNbr[i] : -1;
            d.patches.push_back(p);
        }
        writePolyMesh(dir, location, d, binary != 0, precision);
    });
}
int smhost_set_write_compression(int32_t on) {
    return guarded([&] { setWriteCompression(on != 0); });
}

int smhost_write_points(const char* dir, const char* location, int32_t nPoints, const double* points, int32_t binary, int32_t precision) {
    return guarded([&] { writePoints(dir, location, nPoints, points, binary != 0, precision); });
}
int smhost_read_label_list(const char* file, int32_t* out, int64_t* n) {
    return guarded([&] {
        std::vector<int32_t> v;
        readLabelList(file, v);
        if (out && *n >= (int64_t)v.size()) std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
        *n = (int64_t)v.size();
    });
}
int smhost_read_points(const char* file, double* out, int64_t* n) {
    return guarded([&] {
        std::vector<double> v;
        readPoints(file, v);
        if (out && *n >= (int64_t)v.size()) std::memcpy(out, v.data(), v.size() * sizeof(double));
        *n = (int64_t)v.size();
    });
}
int smhost_read_obj(const char* file, int32_t kind, double* points, int64_t* nPoints, int32_t* elements, int64_t* nElements) {
    return guarded([&] {
        std::vector<double> p;
        std::vector<int32_t> e;
        if (kind == 0) readObjSurface(file, p, e);
        else if (kind == 1) readObjEdges(file, p, e);
        else throw std::runtime_error("smhost_read_obj: kind must be 0 (surface) or 1 (edge mesh)");
        const int64_t w = kind == 0 ? 3 : 2;
        if (points && elements && *nPoints >= (int64_t)p.size() / 3 && *nElements >= (int64_t)e.size() / w) {
            std::memcpy(points, p.data(), p.size() * sizeof(double));
            std::memcpy(elements, e.data(), e.size() * sizeof(int32_t));
        }
        *nPoints = (int64_t)p.size() / 3;
        *nElements = (int64_t)e.size() / w;
    });
}
int smhost_write_label_list(const char* file, const char* location, const char* object, const char* cls, int64_t n,
                            const int32_t* values, int32_t binary) {
    return guarded([&] { writeLabelList(file, location, object, cls, n, values, binary != 0); });
}
int smhost_gen_cavity_mesh(int32_t N, double radius, double shell, double jitter, uint64_t seed, smhost_mesh** out) {
    return guarded([&] {
        auto* m = new smhost_mesh();
        try { genCavityMesh(N, radius, shell, jitter, seed, m->d); }
        catch (...) { delete m; throw; }
        *out = m;
    });
}
int smhost_gen_cavity_subdomain(int32_t N, double radius, double shell, double jitter, uint64_t seed, int32_t px, int32_t py, int32_t pz,
                                int32_t rank, smhost_mesh** out) {
    return guarded([&] {
        auto* m = new smhost_mesh();
        const int grid[3] = {px, py, pz};
        try { genCavitySubdomain(N, radius, shell, jitter, seed, grid, rank, m->d, &m->pointGlobal, &m->cellGlobal); }
        catch (...) { delete m; throw; }
        *out = m;
    });
}
int smhost_mesh_global_ids(const smhost_mesh* m, int64_t* pointGlobal, int64_t* cellGlobal) {
    if ((int64_t)m->pointGlobal.size() != m->d.nPoints() || (int64_t)m->cellGlobal.size() != m->d.nCells) {
        g_err = "smhost_mesh_global_ids: this mesh carries no global ids (only smhost_gen_cavity_subdomain meshes do)";
        return 1;
    }
    if (pointGlobal) std::memcpy(pointGlobal, m->pointGlobal.data(), m->pointGlobal.size() * sizeof(int64_t));
    if (cellGlobal) std::memcpy(cellGlobal, m->cellGlobal.data(), m->cellGlobal.size() * sizeof(int64_t));
    return 0;
}
}
