// meshgen.cpp -- synthetic polyhedral mesh standing in for a snappyHexMesh "cube with a spherical
// cavity" case (BASELINE configs[3-4]; snappyHexMesh and decomposePar are not available here).
//
// Castellated octree mesh, one refinement level: an N^3 hex grid on the unit cube; coarse cells whose
// centre lies within `shell` of the sphere surface are split into 8; every leaf (coarse or fine)
// whose centre is inside the sphere is removed.  Coarse cells next to refined ones become genuinely
// polyhedral, exactly as in a snappyHexMesh castellated mesh: their faces towards the refined side
// are split in 4, and faces that merely touch a refined edge carry the hanging mid-edge point
// (5..8-vertex polygons).  Output is a valid polyMesh in OpenFOAM ordering: leaves numbered
// lexicographically (children in place of their parent), internal faces upper-triangular (owner
// ascending, then neighbour), normals owner -> neighbour, boundary patches xmin..zmax + "cavity".
//
// The same routine generates ONE sub-domain of a (Px, Py, Pz) box decomposition of that mesh directly, in
// decomposePar layout (the reference runs on decomposePar output, testcase/run_parallel:11-19, SM.C:49-58):
// coarse cell i of an axis belongs to box b with floor(b N / P) <= i < floor((b + 1) N / P), children follow their
// parent; local cells / points / faces keep ascending global order; faces = internal, the seven physical patches
// (kept when empty), one processor patch per neighbouring rank (ascending rank, faces in ascending global face
// order); a processor face whose local cell is the global neighbour is reversed about its first vertex.  Only the
// box and a halo of one coarse cell (for the "is this point on a boundary patch" test of the jitter) are ever
// visited, so a rank of an 8-way 80 M-cell case holds an eighth of the mesh.  The result equals
// decompose(global mesh, that partition)[rank] array for array (tests/test_host_logic.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
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

// pointGlobal (may be NULL): per local point its index in the global fine lattice, (2N+1)^3, x fastest -- ascending
// with the global point id, unique per global point: serves as pointProcAddressing for the shared-point tables.
// cellGlobal (may be NULL): per local cell 8 * (coarse cell index, x fastest) + child -- ascending with the global cell id.
void genCavitySubdomain(int N, double radius, double shell, double jitter, uint64_t seed, const int grid[3], int rank,
                        PolyMeshData& out, std::vector<int64_t>* pointGlobal, std::vector<int64_t>* cellGlobal) {
    if (N < 2 || N > 1000) throw std::runtime_error("genCavityMesh: N out of range");
    const int Px = grid[0], Py = grid[1], Pz = grid[2];
    if (Px < 1 || Py < 1 || Pz < 1 || Px > N || Py > N || Pz > N) throw std::runtime_error("genCavityMesh: bad processor grid");
    const int nRanks = Px * Py * Pz;
    if (rank < 0 || rank >= nRanks) throw std::runtime_error("genCavityMesh: rank out of range");
    const int M = 2 * N;          // fine cells per side
    const int64_t L = M + 1;      // fine lattice points per side
    const double hc = 1.0 / N, hf = 0.5 * hc;
    const int P3[3] = {Px, Py, Pz};
    const int r3[3] = {rank % Px, (rank / Px) % Py, rank / (Px * Py)};
    int c0[3], c1[3], e0[3], e1[3];   // my coarse range, and the range with the one-cell halo
    for (int a = 0; a < 3; ++a) {
        c0[a] = (int)(((int64_t)r3[a] * N) / P3[a]);
        c1[a] = (int)(((int64_t)(r3[a] + 1) * N) / P3[a]);
        e0[a] = std::max(c0[a] - 1, 0);
        e1[a] = std::min(c1[a] + 1, N);
    }
    auto boxOf = [&](int a, int i) {   // box of coarse index i along axis a
        int b = (int)(((int64_t)i * P3[a]) / N);
        while (b + 1 < P3[a] && (int)(((int64_t)(b + 1) * N) / P3[a]) <= i) ++b;
        while (b > 0 && (int)(((int64_t)b * N) / P3[a]) > i) --b;
        return b;
    };
    auto rankOfCoarse = [&](int i, int j, int k) { return boxOf(0, i) + Px * (boxOf(1, j) + Py * boxOf(2, k)); };
    auto dist = [&](double x, double y, double z) { return std::sqrt((x - 0.5) * (x - 0.5) + (y - 0.5) * (y - 0.5) + (z - 0.5) * (z - 0.5)); };
    auto gpidx = [&](int i, int j, int k) { return (int64_t)i + L * ((int64_t)j + L * (int64_t)k); };

    // refinement flags, cached over the halo range widened by one more cell (edge tests of halo faces), analytic elsewhere
    int q0[3], q1[3];
    for (int a = 0; a < 3; ++a) { q0[a] = std::max(e0[a] - 1, 0); q1[a] = std::min(e1[a] + 1, N); }
    const int qn[3] = {q1[0] - q0[0], q1[1] - q0[1], q1[2] - q0[2]};
    auto refinedAnalytic = [&](int i, int j, int k) -> bool {
        const double d = dist((i + 0.5) * hc, (j + 0.5) * hc, (k + 0.5) * hc);
        return std::fabs(d - radius) <= shell;
    };
    std::vector<uint8_t> refinedCache((size_t)qn[0] * qn[1] * qn[2]);
    for (int k = q0[2]; k < q1[2]; ++k) for (int j = q0[1]; j < q1[1]; ++j) for (int i = q0[0]; i < q1[0]; ++i)
        refinedCache[(size_t)(i - q0[0]) + (size_t)qn[0] * ((size_t)(j - q0[1]) + (size_t)qn[1] * (size_t)(k - q0[2]))] = refinedAnalytic(i, j, k) ? 1 : 0;
    auto isRefinedCell = [&](int i, int j, int k) -> bool {   // false outside the domain
        if (i < 0 || j < 0 || k < 0 || i >= N || j >= N || k >= N) return false;
        if (i >= q0[0] && i < q1[0] && j >= q0[1] && j < q1[1] && k >= q0[2] && k < q1[2])
            return refinedCache[(size_t)(i - q0[0]) + (size_t)qn[0] * ((size_t)(j - q0[1]) + (size_t)qn[1] * (size_t)(k - q0[2]))] != 0;
        return refinedAnalytic(i, j, k);
    };

    // leaves over the halo range: local ids for my own leaves (ascending = ascending global id), -1 removed, -3 foreign
    const int en[3] = {e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2]};
    auto ecidx = [&](int i, int j, int k) { return (size_t)(i - e0[0]) + (size_t)en[0] * ((size_t)(j - e0[1]) + (size_t)en[1] * (size_t)(k - e0[2])); };
    const int fn[3] = {2 * en[0], 2 * en[1], 2 * en[2]};
    auto efidx = [&](int fi, int fj, int fk) {
        return (size_t)(fi - 2 * e0[0]) + (size_t)fn[0] * ((size_t)(fj - 2 * e0[1]) + (size_t)fn[1] * (size_t)(fk - 2 * e0[2]));
    };
    std::vector<int32_t> coarseLeaf((size_t)en[0] * en[1] * en[2], -1);
    std::vector<int32_t> fineLeaf((size_t)fn[0] * fn[1] * fn[2], -1);
    auto mineCoarse = [&](int i, int j, int k) { return i >= c0[0] && i < c1[0] && j >= c0[1] && j < c1[1] && k >= c0[2] && k < c1[2]; };
    int32_t nCells = 0;
    std::vector<int64_t> cellKeys;
    for (int k = e0[2]; k < e1[2]; ++k) for (int j = e0[1]; j < e1[1]; ++j) for (int i = e0[0]; i < e1[0]; ++i) {
        const bool mine = mineCoarse(i, j, k);
        const int64_t ckey = 8 * ((int64_t)i + (int64_t)N * ((int64_t)j + (int64_t)N * (int64_t)k));
        if (!isRefinedCell(i, j, k)) {
            if (dist((i + 0.5) * hc, (j + 0.5) * hc, (k + 0.5) * hc) >= radius) {
                coarseLeaf[ecidx(i, j, k)] = mine ? nCells++ : -3;
                if (mine) cellKeys.push_back(ckey);
            }
        } else {
            for (int c = 0; c < 2; ++c) for (int b = 0; b < 2; ++b) for (int a = 0; a < 2; ++a) {
                const int fi = 2 * i + a, fj = 2 * j + b, fk = 2 * k + c;
                if (dist((fi + 0.5) * hf, (fj + 0.5) * hf, (fk + 0.5) * hf) >= radius) {
                    fineLeaf[efidx(fi, fj, fk)] = mine ? nCells++ : -3;
                    if (mine) cellKeys.push_back(ckey + a + 2 * b + 4 * c);
                }
            }
        }
    }
    if (nCells == 0) throw std::runtime_error("genCavityMesh: no cells left" + std::string(nRanks > 1 ? " in this sub-domain" : ""));

    // leaf covering a fine cell position inside the halo range: local id (>= 0), -1 removed, -2 outside the domain,
    // -3 a leaf of another rank
    auto leafAt = [&](int fi, int fj, int fk) -> int32_t {
        if (fi < 0 || fj < 0 || fk < 0 || fi >= M || fj >= M || fk >= M) return -2;
        const int ci = fi >> 1, cj = fj >> 1, ck = fk >> 1;
        if (ci < e0[0] || ci >= e1[0] || cj < e0[1] || cj >= e1[1] || ck < e0[2] || ck >= e1[2]) return -4;   // beyond the halo (never asked for my leaves)
        if (isRefinedCell(ci, cj, ck)) return fineLeaf[efidx(fi, fj, fk)];
        return coarseLeaf[ecidx(ci, cj, ck)];
    };
    auto keyAt = [&](int fi, int fj, int fk) -> int64_t {     // global order key of the leaf covering a fine cell
        const int ci = fi >> 1, cj = fj >> 1, ck = fk >> 1;
        const int64_t ckey = 8 * ((int64_t)ci + (int64_t)N * ((int64_t)cj + (int64_t)N * (int64_t)ck));
        return isRefinedCell(ci, cj, ck) ? ckey + (fi & 1) + 2 * (fj & 1) + 4 * (fk & 1) : ckey;
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

    // patch: -1 internal, 0..6 physical, 7 + r = processor patch towards rank r
    struct Face { int32_t own, nei; int32_t patch; int32_t off, n; int64_t k0, k1; };
    std::vector<Face> faces;
    std::vector<int64_t> verts;   // global lattice indices, remapped to point ids at the end
    faces.reserve((size_t)nCells * 4);
    verts.reserve((size_t)nCells * 16);
    // lattice points of my box that lie on a face of a physical patch (of any rank's leaf): they are not jittered
    const int64_t b0[3] = {2 * (int64_t)c0[0], 2 * (int64_t)c0[1], 2 * (int64_t)c0[2]};
    const int64_t bn[3] = {2 * (int64_t)(c1[0] - c0[0]) + 1, 2 * (int64_t)(c1[1] - c0[1]) + 1, 2 * (int64_t)(c1[2] - c0[2]) + 1};
    auto inBox = [&](int x, int y, int z) {
        return x >= b0[0] && x < b0[0] + bn[0] && y >= b0[1] && y < b0[1] + bn[1] && z >= b0[2] && z < b0[2] + bn[2];
    };
    auto bidx = [&](int x, int y, int z) { return (size_t)(x - b0[0]) + (size_t)bn[0] * ((size_t)(y - b0[1]) + (size_t)bn[1] * (size_t)(z - b0[2])); };
    std::vector<int32_t> pid((size_t)(bn[0] * bn[1] * bn[2]), -1);   // -1 unused, >= 0 point id (0 = used, before numbering)
    std::vector<uint8_t> onBoundary((size_t)(bn[0] * bn[1] * bn[2]), 0);

    // the vertex loop of a face in the plane `ax = pos` (lattice units) spanning [u0,u0+size] x [v0,v0+size], normal
    // +ax; coarse faces (size 2) get hanging mid-edge points where the edge is split
    auto faceLoop = [&](int ax, int pos, int u0, int v0, int size, int loop[8][3]) -> int {
        const int u = (ax + 1) % 3, v = (ax + 2) % 3;
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
        return n;
    };
    // a face of one of my leaves; positive: normal +ax (vertex loop as it stands), else the loop reversed about its
    // first vertex
    auto emit = [&](int ax, int pos, int u0, int v0, int size, bool positive, int32_t own, int32_t nei, int32_t patch, int64_t k0, int64_t k1) {
        int loop[8][3];
        const int n = faceLoop(ax, pos, u0, v0, size, loop);
        Face f{own, nei, patch, (int32_t)verts.size(), n, k0, k1};
        if (positive) for (int i = 0; i < n; ++i) verts.push_back(gpidx(loop[i][0], loop[i][1], loop[i][2]));
        else { verts.push_back(gpidx(loop[0][0], loop[0][1], loop[0][2])); for (int i = n - 1; i >= 1; --i) verts.push_back(gpidx(loop[i][0], loop[i][1], loop[i][2])); }
        for (int i = 0; i < n; ++i) {
            pid[bidx(loop[i][0], loop[i][1], loop[i][2])] = 0;
            if (patch >= 0 && patch < 7) onBoundary[bidx(loop[i][0], loop[i][1], loop[i][2])] = 1;
        }
        faces.push_back(f);
    };
    // a physical-patch face of a halo leaf: only its vertices inside my box matter
    auto markBoundary = [&](int ax, int pos, int u0, int v0, int size) {
        int loop[8][3];
        const int n = faceLoop(ax, pos, u0, v0, size, loop);
        for (int i = 0; i < n; ++i)
            if (inBox(loop[i][0], loop[i][1], loop[i][2])) onBoundary[bidx(loop[i][0], loop[i][1], loop[i][2])] = 1;
    };

    const int PATCH_CAVITY = 6;
    // visit leaves in id order (halo leaves in between only mark boundary points)
    for (int k = e0[2]; k < e1[2]; ++k) for (int j = e0[1]; j < e1[1]; ++j) for (int i = e0[0]; i < e1[0]; ++i) {
        const bool ref = isRefinedCell(i, j, k);
        const bool mine = mineCoarse(i, j, k);
        const int nsub = ref ? 2 : 1;
        for (int c = 0; c < nsub; ++c) for (int b = 0; b < nsub; ++b) for (int a = 0; a < nsub; ++a) {
            int f0[3];   // fine-cell origin of the leaf and its size in fine cells
            int size;
            int32_t me;
            if (ref) { f0[0] = 2 * i + a; f0[1] = 2 * j + b; f0[2] = 2 * k + c; size = 1; me = fineLeaf[efidx(f0[0], f0[1], f0[2])]; }
            else { f0[0] = 2 * i; f0[1] = 2 * j; f0[2] = 2 * k; size = 2; me = coarseLeaf[ecidx(i, j, k)]; }
            if (me == -1) continue;
            const int64_t myKey = keyAt(f0[0], f0[1], f0[2]);
            for (int ax = 0; ax < 3; ++ax) {
                const int u = (ax + 1) % 3, v = (ax + 2) % 3;
                for (int side = 0; side < 2; ++side) {
                    const bool plus = side == 1;
                    const int pos = f0[ax] + (plus ? size : 0);            // lattice plane of the face
                    int nb[3] = {f0[0], f0[1], f0[2]};
                    nb[ax] = plus ? f0[ax] + size : f0[ax] - 1;            // a fine cell just across the face
                    const bool outside = nb[ax] < 0 || nb[ax] >= M;
                    if (!mine) {
                        // halo leaf: its faces on physical patches (domain boundary / removed neighbour) mark points
                        if (outside) { markBoundary(ax, pos, f0[u], f0[v], size); continue; }
                        // a neighbour beyond the halo can only hide cavity faces whose points are out of my box anyway
                        const int nci = nb[0] >> 1, ncj = nb[1] >> 1, nck = nb[2] >> 1;
                        if (nci < e0[0] || nci >= e1[0] || ncj < e0[1] || ncj >= e1[1] || nck < e0[2] || nck >= e1[2]) {
                            // still exact: decide from the analytic description
                            const bool nbRef = isRefinedCell(nci, ncj, nck);
                            if (size == 2 && nbRef) {
                                for (int dv = 0; dv < 2; ++dv) for (int du = 0; du < 2; ++du) {
                                    int q[3] = {nb[0], nb[1], nb[2]};
                                    q[u] = f0[u] + du; q[v] = f0[v] + dv;
                                    if (dist((q[0] + 0.5) * hf, (q[1] + 0.5) * hf, (q[2] + 0.5) * hf) < radius) markBoundary(ax, pos, f0[u] + du, f0[v] + dv, 1);
                                }
                            } else {
                                const bool removed = nbRef ? dist((nb[0] + 0.5) * hf, (nb[1] + 0.5) * hf, (nb[2] + 0.5) * hf) < radius
                                                           : dist((nci + 0.5) * hc, (ncj + 0.5) * hc, (nck + 0.5) * hc) < radius;
                                if (removed) markBoundary(ax, pos, f0[u], f0[v], size);
                            }
                            continue;
                        }
                        const bool nbRefined = isRefinedCell(nci, ncj, nck);
                        if (size == 2 && nbRefined) {
                            for (int dv = 0; dv < 2; ++dv) for (int du = 0; du < 2; ++du) {
                                int q[3] = {nb[0], nb[1], nb[2]};
                                q[u] = f0[u] + du; q[v] = f0[v] + dv;
                                if (leafAt(q[0], q[1], q[2]) == -1) markBoundary(ax, pos, f0[u] + du, f0[v] + dv, 1);
                            }
                        } else if (leafAt(nb[0], nb[1], nb[2]) == -1) markBoundary(ax, pos, f0[u], f0[v], size);
                        continue;
                    }
                    if (outside) {   // domain boundary
                        emit(ax, pos, f0[u], f0[v], size, plus, me, -1, 2 * ax + side, 0, 0);
                        continue;
                    }
                    const int32_t first = leafAt(nb[0], nb[1], nb[2]);
                    const bool nbRefined = isRefinedCell(nb[0] >> 1, nb[1] >> 1, nb[2] >> 1);
                    const int nbRank = rankOfCoarse(nb[0] >> 1, nb[1] >> 1, nb[2] >> 1);
                    // the global mesh holds the face once, emitted by the lower leaf on its plus side with the +ax loop;
                    // towards another rank it is a processor face: as it stands when I am the global owner (plus side),
                    // reversed about its first vertex when I am the global neighbour (minus side)
                    auto other = [&](int32_t o, int q0_, int q1_, int q2_, int u0, int v0, int sz) {
                        if (o == -1) emit(ax, pos, u0, v0, sz, plus, me, -1, PATCH_CAVITY, 0, 0);
                        else if (o == -3) {
                            const int64_t ok = keyAt(q0_, q1_, q2_);
                            emit(ax, pos, u0, v0, sz, plus, me, -1, 7 + nbRank, plus ? myKey : ok, plus ? ok : myKey);
                        } else if (plus) emit(ax, pos, u0, v0, sz, true, me, o, -1, 0, 0);
                    };
                    if (size == 2 && nbRefined) {
                        // coarse leaf against a refined parent: four fine sub-faces
                        for (int dv = 0; dv < 2; ++dv) for (int du = 0; du < 2; ++du) {
                            int q[3] = {nb[0], nb[1], nb[2]};
                            q[u] = f0[u] + du; q[v] = f0[v] + dv;
                            other(leafAt(q[0], q[1], q[2]), q[0], q[1], q[2], f0[u] + du, f0[v] + dv, 1);
                        }
                    } else {
                        // same-size neighbour, or fine leaf against a coarse leaf: one face of this leaf's size
                        other(first, nb[0], nb[1], nb[2], f0[u], f0[v], size);
                    }
                }
            }
        }
    }

    // order: internal faces by (owner, neighbour), then the physical patches (owner order kept), then the processor
    // patches by rank with their faces in global face order = (global owner, global neighbour)
    std::vector<int32_t> order(faces.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const Face &x = faces[a], &y = faces[b];
        if ((x.patch < 0) != (y.patch < 0)) return x.patch < 0;
        if (x.patch < 0) return x.own != y.own ? x.own < y.own : x.nei < y.nei;
        if (x.patch != y.patch) return x.patch < y.patch;
        if (x.patch >= 7) return x.k0 != y.k0 ? x.k0 < y.k0 : x.k1 < y.k1;
        return false;   // stable: owner order kept inside a physical patch
    });
    for (size_t i = 0; i < order.size(); ++i)
        if (faces[order[i]].patch < 0 && faces[order[i]].own >= faces[order[i]].nei) throw std::runtime_error("genCavityMesh: owner >= neighbour");

    // points: ids ascending in lattice order
    int32_t nPoints = 0;
    for (size_t i = 0; i < pid.size(); ++i) if (pid[i] == 0) pid[i] = nPoints++;
    out = PolyMeshData();
    out.nCells = nCells;
    out.points.assign((size_t)nPoints * 3, 0.0);
    if (pointGlobal) pointGlobal->assign((size_t)nPoints, 0);
    for (int64_t z = b0[2]; z < b0[2] + bn[2]; ++z) for (int64_t y = b0[1]; y < b0[1] + bn[1]; ++y) for (int64_t x = b0[0]; x < b0[0] + bn[0]; ++x) {
        const size_t bi = bidx((int)x, (int)y, (int)z);
        const int32_t id = pid[bi];
        if (id < 0) continue;
        double c[3] = {x * hf, y * hf, z * hf};
        const uint64_t g = (uint64_t)gpidx((int)x, (int)y, (int)z);
        if (jitter > 0.0 && !onBoundary[bi]) {
            for (int a = 0; a < 3; ++a) {
                const double r = (double)(splitmix(seed * 0x9E3779B97F4A7C15ull + g * 3 + a) >> 11) * (1.0 / 9007199254740992.0);
                c[a] += (2.0 * r - 1.0) * jitter * hf;
            }
        }
        out.points[3 * (size_t)id] = c[0]; out.points[3 * (size_t)id + 1] = c[1]; out.points[3 * (size_t)id + 2] = c[2];
        if (pointGlobal) (*pointGlobal)[(size_t)id] = (int64_t)g;
    }
    if (cellGlobal) *cellGlobal = cellKeys;
    out.faceOffsets.assign(1, 0);
    const char* names[7] = {"xmin", "xmax", "ymin", "ymax", "zmin", "zmax", "cavity"};
    std::map<int32_t, int32_t> patchCount;
    for (int p = 0; p < 7; ++p) patchCount[p] = 0;
    auto localPid = [&](int64_t g) {
        const int x = (int)(g % L), y = (int)((g / L) % L), z = (int)(g / (L * L));
        return pid[bidx(x, y, z)];
    };
    for (int32_t fi : order) {
        const Face& f = faces[fi];
        for (int32_t t = f.off; t < f.off + f.n; ++t) out.facePoints.push_back(localPid(verts[t]));
        out.faceOffsets.push_back((int32_t)out.facePoints.size());
        out.owner.push_back(f.own);
        if (f.patch < 0) out.neighbour.push_back(f.nei);
        else patchCount[f.patch]++;
    }
    int32_t start = (int32_t)out.neighbour.size();
    for (const auto& pc : patchCount) {
        PatchInfo pi;
        if (pc.first < 7) { pi.name = names[pc.first]; pi.type = (pc.first == 6) ? "wall" : "patch"; }
        else {
            pi.name = "procBoundary" + std::to_string(rank) + "to" + std::to_string(pc.first - 7);
            pi.type = "processor"; pi.myProcNo = rank; pi.neighbProcNo = pc.first - 7;
        }
        pi.nFaces = pc.second; pi.startFace = start;
        start += pc.second;
        out.patches.push_back(pi);
    }
}

void genCavityMesh(int N, double radius, double shell, double jitter, uint64_t seed, PolyMeshData& out) {
    const int grid[3] = {1, 1, 1};
    if (N > 500) throw std::runtime_error("genCavityMesh: N out of range");      // (32-bit face-vertex offsets: 4 x 3 x N^3 entries)
    genCavitySubdomain(N, radius, shell, jitter, seed, grid, 0, out, nullptr, nullptr);
}

}  // namespace smhost
