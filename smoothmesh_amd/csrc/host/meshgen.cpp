// meshgen.cpp -- synthetic polyhedral mesh standing in for a snappyHexMesh "cube with a spherical
// cavity" case (BASELINE configs[3-4]; snappyHexMesh is not available here).
//
// Castellated octree mesh, one refinement level: an N^3 hex grid on the unit cube; coarse cells whose
// centre lies within `shell` of the sphere surface are split into 8; every leaf (coarse or fine)
// whose centre is inside the sphere is removed.  Coarse cells next to refined ones become genuinely
// polyhedral, exactly as in a snappyHexMesh castellated mesh: their faces towards the refined side
// are split in 4, and faces that merely touch a refined edge carry the hanging mid-edge point
// (5..8-vertex polygons).  Output is a valid polyMesh in OpenFOAM ordering: leaves numbered
// lexicographically (children in place of their parent), internal faces upper-triangular (owner
// ascending, then neighbour), normals owner -> neighbour, boundary patches xmin..zmax + "cavity".
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "polymesh_io.hpp"

namespace smhost {

namespace {
inline uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
}  // namespace

void genCavityMesh(int N, double radius, double shell, double jitter, uint64_t seed, PolyMeshData& out) {
    if (N < 2 || N > 400) throw std::runtime_error("genCavityMesh: N out of range");
    const int M = 2 * N;          // fine cells per side
    const int L = M + 1;          // fine lattice points per side
    const double hc = 1.0 / N, hf = 0.5 * hc;
    auto cidx = [&](int i, int j, int k) { return (size_t)i + (size_t)N * ((size_t)j + (size_t)N * (size_t)k); };
    auto fidx = [&](int i, int j, int k) { return (size_t)i + (size_t)M * ((size_t)j + (size_t)M * (size_t)k); };
    auto pidx = [&](int i, int j, int k) { return (size_t)i + (size_t)L * ((size_t)j + (size_t)L * (size_t)k); };
    auto dist = [&](double x, double y, double z) { return std::sqrt((x - 0.5) * (x - 0.5) + (y - 0.5) * (y - 0.5) + (z - 0.5) * (z - 0.5)); };

    std::vector<uint8_t> refined((size_t)N * N * N, 0);
    for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
        const double d = dist((i + 0.5) * hc, (j + 0.5) * hc, (k + 0.5) * hc);
        refined[cidx(i, j, k)] = (std::fabs(d - radius) <= shell) ? 1 : 0;
    }
    // leaf numbering
    std::vector<int32_t> coarseLeaf((size_t)N * N * N, -1);
    std::vector<int32_t> fineLeaf((size_t)M * M * M, -1);   // only children of refined parents are used
    int32_t nCells = 0;
    for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
        if (!refined[cidx(i, j, k)]) {
            if (dist((i + 0.5) * hc, (j + 0.5) * hc, (k + 0.5) * hc) >= radius) coarseLeaf[cidx(i, j, k)] = nCells++;
        } else {
            for (int c = 0; c < 2; ++c) for (int b = 0; b < 2; ++b) for (int a = 0; a < 2; ++a) {
                const int fi = 2 * i + a, fj = 2 * j + b, fk = 2 * k + c;
                if (dist((fi + 0.5) * hf, (fj + 0.5) * hf, (fk + 0.5) * hf) >= radius) fineLeaf[fidx(fi, fj, fk)] = nCells++;
            }
        }
    }
    if (nCells == 0) throw std::runtime_error("genCavityMesh: no cells left");

    // leaf covering a fine cell position: returns id, sets isCoarse; -1 = removed, -2 = outside
    auto leafAt = [&](int fi, int fj, int fk, bool& isCoarse) -> int32_t {
        if (fi < 0 || fj < 0 || fk < 0 || fi >= M || fj >= M || fk >= M) return -2;
        const size_t c = cidx(fi >> 1, fj >> 1, fk >> 1);
        if (refined[c]) { isCoarse = false; return fineLeaf[fidx(fi, fj, fk)]; }
        isCoarse = true;
        return coarseLeaf[c];
    };
    auto isRefinedCell = [&](int i, int j, int k) -> bool {
        if (i < 0 || j < 0 || k < 0 || i >= N || j >= N || k >= N) return false;
        return refined[cidx(i, j, k)] != 0;
    };
    // coarse edge from even lattice point (x,y,z) along axis ax: split iff a cell around it is refined
    auto edgeSplit = [&](int x, int y, int z, int ax) -> bool {
        int c[3] = {x >> 1, y >> 1, z >> 1};
        const int u = (ax + 1) % 3, v = (ax + 2) % 3;
        for (int du = -1; du <= 0; ++du) for (int dv = -1; dv <= 0; ++dv) {
            int q[3] = {c[0], c[1], c[2]};
            q[u] += du; q[v] += dv;
            if (isRefinedCell(q[0], q[1], q[2])) return true;
        }
        return false;
    };

    struct Face { int32_t own, nei; int32_t patch; int32_t off, n; };   // patch -1 = internal
    std::vector<Face> faces;
    std::vector<int64_t> verts;   // lattice indices, remapped to point ids at the end
    faces.reserve((size_t)nCells * 4);
    verts.reserve((size_t)nCells * 16);

    // emit a face in the plane `ax = pos` (lattice units) spanning [u0,u0+size] x [v0,v0+size]; normal +ax if
    // positive else -ax.  coarse faces (size 2) get hanging mid-edge points where the edge is split.
    auto emit = [&](int ax, int pos, int u0, int v0, int size, bool positive, int32_t own, int32_t nei, int32_t patch) {
        const int u = (ax + 1) % 3, v = (ax + 2) % 3;
        int loop[8][3];
        int n = 0;
        const int cu[4] = {0, 1, 1, 0}, cv[4] = {0, 0, 1, 1};
        for (int e = 0; e < 4; ++e) {
            int p[3]; p[ax] = pos; p[u] = u0 + cu[e] * size; p[v] = v0 + cv[e] * size;
            loop[n][0] = p[0]; loop[n][1] = p[1]; loop[n][2] = p[2]; ++n;
            if (size == 2) {
                const int e2 = (e + 1) & 3;
                int q[3]; q[ax] = pos; q[u] = u0 + cu[e2] * size; q[v] = v0 + cv[e2] * size;
                // the edge p -> q runs along u (e = 0, 2) or v (e = 1, 3)
                const int eax = (e & 1) ? v : u;
                int lo[3] = {std::min(p[0], q[0]), std::min(p[1], q[1]), std::min(p[2], q[2])};
                if (edgeSplit(lo[0], lo[1], lo[2], eax)) {
                    loop[n][0] = (p[0] + q[0]) / 2; loop[n][1] = (p[1] + q[1]) / 2; loop[n][2] = (p[2] + q[2]) / 2; ++n;
                }
            }
        }
        Face f{own, nei, patch, (int32_t)verts.size(), n};
        if (positive) for (int i = 0; i < n; ++i) verts.push_back((int64_t)pidx(loop[i][0], loop[i][1], loop[i][2]));
        else { verts.push_back((int64_t)pidx(loop[0][0], loop[0][1], loop[0][2])); for (int i = n - 1; i >= 1; --i) verts.push_back((int64_t)pidx(loop[i][0], loop[i][1], loop[i][2])); }
        faces.push_back(f);
    };

    const int PATCH_CAVITY = 6;
    // visit leaves in id order
    for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
        const bool ref = refined[cidx(i, j, k)] != 0;
        const int nsub = ref ? 2 : 1;
        for (int c = 0; c < nsub; ++c) for (int b = 0; b < nsub; ++b) for (int a = 0; a < nsub; ++a) {
            int f0[3];   // fine-cell origin of the leaf and its size in fine cells
            int size;
            int32_t me;
            if (ref) { f0[0] = 2 * i + a; f0[1] = 2 * j + b; f0[2] = 2 * k + c; size = 1; me = fineLeaf[fidx(f0[0], f0[1], f0[2])]; }
            else { f0[0] = 2 * i; f0[1] = 2 * j; f0[2] = 2 * k; size = 2; me = coarseLeaf[cidx(i, j, k)]; }
            if (me < 0) continue;
            for (int ax = 0; ax < 3; ++ax) {
                const int u = (ax + 1) % 3, v = (ax + 2) % 3;
                for (int side = 0; side < 2; ++side) {
                    const bool plus = side == 1;
                    const int pos = f0[ax] + (plus ? size : 0);            // lattice plane of the face
                    int nb[3] = {f0[0], f0[1], f0[2]};
                    nb[ax] = plus ? f0[ax] + size : f0[ax] - 1;            // a fine cell just across the face
                    bool nbCoarse = false;
                    const int32_t first = leafAt(nb[0], nb[1], nb[2], nbCoarse);
                    if (first == -2) {   // domain boundary
                        emit(ax, pos, f0[u], f0[v], size, plus, me, -1, 2 * ax + side);
                        continue;
                    }
                    const bool nbRefined = refined[cidx(nb[0] >> 1, nb[1] >> 1, nb[2] >> 1)] != 0;
                    if (size == 2 && nbRefined) {
                        // coarse leaf against a refined parent: four fine sub-faces
                        for (int dv = 0; dv < 2; ++dv) for (int du = 0; du < 2; ++du) {
                            int q[3] = {nb[0], nb[1], nb[2]};
                            q[u] = f0[u] + du; q[v] = f0[v] + dv;
                            bool dummy;
                            const int32_t other = leafAt(q[0], q[1], q[2], dummy);
                            if (other < 0) emit(ax, pos, f0[u] + du, f0[v] + dv, 1, plus, me, -1, PATCH_CAVITY);
                            else if (plus) emit(ax, pos, f0[u] + du, f0[v] + dv, 1, true, me, other, -1);
                        }
                    } else {
                        // same-size neighbour, or fine leaf against a coarse leaf: one face of this leaf's size
                        if (first < 0) emit(ax, pos, f0[u], f0[v], size, plus, me, -1, PATCH_CAVITY);
                        else if (plus) emit(ax, pos, f0[u], f0[v], size, true, me, first, -1);
                    }
                }
            }
        }
    }

    // order: internal faces by (owner, neighbour), then patches
    std::vector<int32_t> order(faces.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const Face &x = faces[a], &y = faces[b];
        if ((x.patch < 0) != (y.patch < 0)) return x.patch < 0;
        if (x.patch < 0) return x.own != y.own ? x.own < y.own : x.nei < y.nei;
        return x.patch < y.patch;   // stable: owner order kept inside a patch
    });
    for (size_t i = 0; i < order.size(); ++i)
        if (faces[order[i]].patch < 0 && faces[order[i]].own >= faces[order[i]].nei) throw std::runtime_error("genCavityMesh: owner >= neighbour");

    // points: ids ascending in lattice order
    std::vector<int32_t> pid((size_t)L * L * L, -1);
    for (int64_t v : verts) pid[(size_t)v] = 0;
    std::vector<uint8_t> onBoundary;
    int32_t nPoints = 0;
    for (size_t i = 0; i < pid.size(); ++i) if (pid[i] == 0) pid[i] = nPoints++;
    onBoundary.assign((size_t)nPoints, 0);
    for (const Face& f : faces) if (f.patch >= 0) for (int32_t t = f.off; t < f.off + f.n; ++t) onBoundary[(size_t)pid[(size_t)verts[t]]] = 1;

    out = PolyMeshData();
    out.nCells = nCells;
    out.points.assign((size_t)nPoints * 3, 0.0);
    for (int z = 0; z < L; ++z) for (int y = 0; y < L; ++y) for (int x = 0; x < L; ++x) {
        const int32_t id = pid[pidx(x, y, z)];
        if (id < 0) continue;
        double c[3] = {x * hf, y * hf, z * hf};
        if (jitter > 0.0 && !onBoundary[(size_t)id]) {
            const uint64_t g = (uint64_t)pidx(x, y, z);
            for (int a = 0; a < 3; ++a) {
                const double r = (double)(splitmix(seed * 0x9E3779B97F4A7C15ull + g * 3 + a) >> 11) * (1.0 / 9007199254740992.0);
                c[a] += (2.0 * r - 1.0) * jitter * hf;
            }
        }
        out.points[3 * (size_t)id] = c[0]; out.points[3 * (size_t)id + 1] = c[1]; out.points[3 * (size_t)id + 2] = c[2];
    }
    out.faceOffsets.assign(1, 0);
    const char* names[7] = {"xmin", "xmax", "ymin", "ymax", "zmin", "zmax", "cavity"};
    std::vector<int32_t> patchCount(7, 0);
    for (int32_t fi : order) {
        const Face& f = faces[fi];
        for (int32_t t = f.off; t < f.off + f.n; ++t) out.facePoints.push_back(pid[(size_t)verts[t]]);
        out.faceOffsets.push_back((int32_t)out.facePoints.size());
        out.owner.push_back(f.own);
        if (f.patch < 0) out.neighbour.push_back(f.nei);
        else patchCount[f.patch]++;
    }
    int32_t start = (int32_t)out.neighbour.size();
    for (int p = 0; p < 7; ++p) {
        PatchInfo pi;
        pi.name = names[p]; pi.type = (p == 6) ? "wall" : "patch"; pi.nFaces = patchCount[p]; pi.startFace = start;
        start += patchCount[p];
        out.patches.push_back(pi);
    }
}

}  // namespace smhost
