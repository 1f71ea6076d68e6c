// boundary.cpp -- see boundary.hpp.  Plain IEEE f64 in the reference's evaluation order (-ffp-contract=off).
#include "boundary.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

namespace smgpu {
namespace {
const double kGreat = 1.0e15, kVGreat = 1.0e300;
const double kRelTol = 1e-4, kAbsTol = 1e-6;   // COM.H:20-21

struct N3 { double x, y, z; };
inline N3 ld(const double* v, int p) { return {v[3 * (size_t)p], v[3 * (size_t)p + 1], v[3 * (size_t)p + 2]}; }
inline N3 sub(const N3& a, const N3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dotOf(const N3& a, const N3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double magOf(const N3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// BPS.C:20-79
std::string edgeMeshSanity(const EdgeMeshHost& em, double meshMinEdgeLength, double meshPerimeter) {
    double shortest = kVGreat;
    double lo[3] = {kVGreat, kVGreat, kVGreat}, hi[3] = {-kVGreat, -kVGreat, -kVGreat};
    for (int e = 0; e < em.nEdges(); ++e) {
        const N3 a = ld(em.pts.data(), em.edges[2 * (size_t)e]), b = ld(em.pts.data(), em.edges[2 * (size_t)e + 1]);
        const double len = magOf(sub(b, a));
        if (len < shortest) shortest = len;
        const double c[2][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}};
        for (int k = 0; k < 2; ++k)
            for (int d = 0; d < 3; ++d) {
                if (c[k][d] < lo[d]) lo[d] = c[k][d];
                if (c[k][d] > hi[d]) hi[d] = c[k][d];
            }
    }
    if (shortest < kRelTol * meshMinEdgeLength)
        return "Minimum edge length in edge mesh " + std::to_string(shortest) + " is too small in comparison to minimum edge length in polyMesh";
    const double perimeter = hi[0] - lo[0] + hi[1] - lo[1] + hi[2] + lo[2];   // "+ bbMinZ", BPS.C:69
    if (std::abs((perimeter / meshPerimeter) - 1.0) > 0.5)
        return "Perimeter (sum of bounding box side lengths) of edge mesh " + std::to_string(perimeter) +
               " is too different in comparison to perimeter of polyMesh " + std::to_string(meshPerimeter);
    return "";
}

// projectPointToEdge BPS.C:89-145
inline void projectToEdge(const N3& pt, const EdgeMeshHost& em, int e, double distanceTolerance, N3& proj, int& edgePoint) {
    edgePoint = -1;
    const int ia = em.edges[2 * (size_t)e], ib = em.edges[2 * (size_t)e + 1];
    const N3 a = ld(em.pts.data(), ia), b = ld(em.pts.data(), ib);
    const double edgeLength = magOf(sub(b, a));
    const N3 c2pt = sub(pt, a), edgeVec = sub(b, a);
    const double s = dotOf(c2pt, edgeVec) / (edgeLength * edgeLength);
    const N3 freeProj = {a.x + s * edgeVec.x, a.y + s * edgeVec.y, a.z + s * edgeVec.z};
    if (s <= kAbsTol) {
        proj = a;
        if (magOf(sub(freeProj, a)) <= distanceTolerance) edgePoint = ia;
    } else if (s >= (1.0 - kAbsTol)) {
        proj = b;
        if (magOf(sub(freeProj, b)) <= distanceTolerance) edgePoint = ib;
    } else proj = freeProj;
}

// findClosestEdgeInfo BPS.C:206-264 without a required string
struct ClosestEdge { N3 proj; int edge, string, edgePoint; };
ClosestEdge closestEdge(const N3& pt, const EdgeMeshHost& em, const std::vector<int32_t>& strings, double distanceTolerance) {
    ClosestEdge r{{kGreat, kGreat, kGreat}, -1, -1, -1};
    double best = kGreat;
    const bool haveStrings = (size_t)em.nEdges() == strings.size();
    for (int e = 0; e < em.nEdges(); ++e) {
        N3 proj;
        int ep;
        projectToEdge(pt, em, e, distanceTolerance, proj, ep);
        const double d = magOf(sub(proj, pt));
        if (d < best) {
            best = d;
            r.proj = proj; r.edge = e; r.edgePoint = ep;
            if (haveStrings) r.string = strings[(size_t)e];
        }
    }
    return r;
}

// findContinuousEdgeMeshEdges BPS.C:446-487
inline void continuousEdges(const EdgeMeshHost& em, int e, int& n1, int& n2) {
    n1 = n2 = -1;
    const std::vector<int32_t>& pa = em.pointEdges[(size_t)em.edges[2 * (size_t)e]];
    if (pa.size() == 2) n1 = (pa[0] == e) ? pa[1] : pa[0];
    const std::vector<int32_t>& pb = em.pointEdges[(size_t)em.edges[2 * (size_t)e + 1]];
    if (pb.size() == 2) n2 = (pb[0] == e) ? pb[1] : pb[0];
}

// findEdgeMeshStrings + stringifyEdgeMeshEdges BPS.C:492-587, the recursion unrolled on an explicit stack in the same
// visiting order.  The reference decides whether to descend into a neighbour EDGE n by pointEdges()[n].size() == 2,
// i.e. by the edge count of the POINT whose id equals the edge id (BPS.C:534,542); kept as written, an id past the
// point list counts as "not 2".
void edgeStrings(const EdgeMeshHost& em, std::vector<int32_t>& strings) {
    const int nE = em.nEdges();
    strings.assign((size_t)nE, -1);
    int nStrings = -1;
    struct Frame { int e, n1, n2, s1, s2, stage; };
    std::vector<Frame> stack;
    auto listOfTwo = [&](int id) { return (size_t)id < em.pointEdges.size() && em.pointEdges[(size_t)id].size() == 2; };
    auto enter = [&](int e) {
        Frame f{e, -1, -1, -1, -1, 0};
        continuousEdges(em, e, f.n1, f.n2);
        const int s0 = strings[(size_t)e];
        if (f.n1 >= 0) f.s1 = strings[(size_t)f.n1];
        if (f.n2 >= 0) f.s2 = strings[(size_t)f.n2];
        const int mx = std::max(std::max(s0, f.s1), f.s2);
        if (mx == -1) strings[(size_t)e] = ++nStrings;
        else if (s0 == -1) strings[(size_t)e] = mx;
        stack.push_back(f);
    };
    for (int e0 = 0; e0 < nE; ++e0) {
        if (strings[(size_t)e0] >= 0) continue;
        enter(e0);
        while (!stack.empty()) {
            Frame& f = stack.back();
            if (f.stage == 0) {
                f.stage = 1;
                if (f.n1 >= 0 && f.s1 == -1 && listOfTwo(f.n1)) { enter(f.n1); continue; }
            }
            if (f.stage == 1) {
                f.stage = 2;
                if (f.n2 >= 0 && f.s2 == -1 && listOfTwo(f.n2)) { enter(f.n2); continue; }
            }
            stack.pop_back();
        }
    }
}
}  // namespace

int32_t edgeMeshStrings(const EdgeMeshHost& em, std::vector<int32_t>& strings) {
    edgeStrings(em, strings);
    int32_t mx = -1;
    for (int32_t v : strings) mx = std::max(mx, v);
    return mx + 1;
}

void EdgeMeshHost::buildPointEdges() {
    pointEdges.assign((size_t)nPoints(), {});
    for (int e = 0; e < nEdges(); ++e) {
        pointEdges[(size_t)edges[2 * (size_t)e]].push_back(e);
        if (edges[2 * (size_t)e + 1] != edges[2 * (size_t)e]) pointEdges[(size_t)edges[2 * (size_t)e + 1]].push_back(e);
    }
}

std::string buildBoundarySetup(const Topology& t, const uint8_t* internal, const double* points, const std::vector<BndPatch>& patches,
                               const BoundaryInputHost& in, BoundarySetup& o) {
    const int P = t.nPoints;
    for (const BndPatch& pp : patches)
        if (pp.start < t.nInternalFaces || pp.size < 0 || pp.start + pp.size > t.nFaces) return "boundary set-up: patch face range outside the boundary faces";
    auto checkMesh = [](const EdgeMeshHost& em) {
        for (int32_t v : em.edges) if (v < 0 || v >= em.nPoints()) return false;
        return true;
    };
    if (!checkMesh(in.initEdges) || !checkMesh(in.targetEdges)) return "boundary set-up: edge mesh point id out of range";
    for (int32_t v : in.surfTris) if (v < 0 || (size_t)v >= in.surfPts.size() / 3) return "boundary set-up: surface point id out of range";

    bool ioHaveData = false;   // SM.C:2066-2077
    if (in.isCornerPointIO) for (int p = 0; p < P && !ioHaveData; ++p) ioHaveData = in.isCornerPointIO[p] == 1;
    if (in.isFeatureEdgePointIO) for (int p = 0; p < P && !ioHaveData; ++p) ioHaveData = in.isFeatureEdgePointIO[p] == 1;
    bool anySmoothing = false;
    for (const BndPatch& pp : patches) anySmoothing = anySmoothing || pp.isSmoothing;
    o.enabled = !in.surfTris.empty() && (in.initEdges.nEdges() > 0 || ioHaveData) && anySmoothing;   // SM.C:2080-2093

    EdgeMeshHost init;
    o.target = EdgeMeshHost();
    o.targetEdgeStrings.clear();
    if (o.enabled) {
        init = in.initEdges;
        o.target = in.targetEdges.nEdges() > 0 ? in.targetEdges : in.initEdges;   // SM.C:2148-2160
        init.buildPointEdges();
        o.target.buildPointEdges();
        std::string err = edgeMeshSanity(init, in.meshMinEdgeLength, in.meshPerimeter);
        if (err.empty()) err = edgeMeshSanity(o.target, in.meshMinEdgeLength, in.meshPerimeter);
        if (!err.empty()) return err;
        edgeStrings(o.target, o.targetEdgeStrings);
    }
    const double tol = in.distanceTolerance;

    o.isConnectedToInternalPoint.assign((size_t)P, 0);
    o.isCornerPoint.assign((size_t)P, 0);
    o.isFeatureEdgePoint.assign((size_t)P, 0);
    o.isSmoothingSurfacePoint.assign((size_t)P, 0);
    o.isFrozenSurfacePoint.assign((size_t)P, 0);
    o.isCornerPointOut.assign((size_t)P, 0);
    o.isFeatureEdgePointOut.assign((size_t)P, 0);
    if (in.isCornerPointIO) o.isCornerPointOut.assign(in.isCornerPointIO, in.isCornerPointIO + P);
    if (in.isFeatureEdgePointIO) o.isFeatureEdgePointOut.assign(in.isFeatureEdgePointIO, in.isFeatureEdgePointIO + P);
    o.cornerPoints.assign(3 * (size_t)P, kGreat);
    o.pointStrings.assign((size_t)P, -1);
    o.nCorner = o.nFeature = o.nSmoothingSurface = o.nFrozenSurface = 0;
    const Csr& fp = t.facePoints;
    const Csr& pe = t.pointEdges;   // pointPoints shares its offsets

    // classifyBoundaryPoints BPS.C:296-420: every boundary point is classified on the first patch that holds it
    const bool haveEdges = init.nPoints() > 0 && o.target.nPoints() > 0;
    // pass 1 (visiting order): the boundary points in the order the reference meets them, each with its first patch
    std::vector<uint8_t> visited((size_t)P, 0);
    std::vector<int32_t> met;
    std::vector<uint8_t> metSmoothing;
    for (const BndPatch& pp : patches)
        for (int f = pp.start; f < pp.start + pp.size; ++f)
            for (int k = fp.off[f]; k < fp.off[f + 1]; ++k) {
                const int p = fp.val[k];
                if (visited[p]) continue;
                visited[p] = 1;
                if (internal[p]) continue;
                met.push_back(p);
                metSmoothing.push_back(pp.isSmoothing ? 1 : 0);
            }
    // pass 2 (independent per point, on the host's threads): neighbours, closest initial edge, closest target corner
    std::vector<int32_t> cornerOf(met.size(), -2);   // -2 not a corner, -1 no eligible corner point, else the target point
    {
        auto work = [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const int p = met[i];
                for (int j = pe.off[p]; j < pe.off[p + 1]; ++j)
                    if (internal[t.pointPoints[j]]) { o.isConnectedToInternalPoint[p] = 1; break; }
                if (!haveEdges) continue;
                const N3 pt = ld(points, p);
                if (ioHaveData) {
                    o.isCornerPoint[p] = o.isCornerPointOut[p] == 1;
                    o.isFeatureEdgePoint[p] = o.isFeatureEdgePointOut[p] == 1;
                } else {
                    const ClosestEdge ce = closestEdge(pt, init, o.targetEdgeStrings, tol);
                    if (ce.edgePoint >= 0 && init.pointEdges[(size_t)ce.edgePoint].size() != 2) {
                        o.isCornerPoint[p] = 1;
                        o.isCornerPointOut[p] = 1;
                    } else if (magOf(sub(pt, ce.proj)) < tol) {
                        o.isFeatureEdgePoint[p] = 1;
                        o.isFeatureEdgePointOut[p] = 1;
                    }
                }
                if (o.isCornerPoint[p]) {
                    // findClosestEdgeMeshCornerPointIndex BPS.C:151-183 on the target edge mesh
                    double best = kGreat;
                    int bestPoint = -1;
                    for (int q = 0; q < o.target.nPoints(); ++q) {
                        if (o.target.pointEdges[(size_t)q].size() == 2) continue;
                        const double d = magOf(sub(pt, ld(o.target.pts.data(), q)));
                        if (d < best) { best = d; bestPoint = q; }
                    }
                    cornerOf[i] = bestPoint;
                }
            }
        };
        const size_t n = met.size();
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const size_t nThreads = (n * (size_t)std::max(1, init.nEdges()) < 2000000) ? 1 : hw;
        if (nThreads <= 1) work(0, n);
        else {
            std::vector<std::thread> pool;
            for (size_t k = 0; k < nThreads; ++k) pool.emplace_back(work, n * k / nThreads, n * (k + 1) / nThreads);
            for (auto& th : pool) th.join();
        }
    }
    // pass 3: counts and corner targets
    for (size_t i = 0; i < met.size(); ++i) {
        const int p = met[i];
        if (haveEdges) {
            if (o.isCornerPoint[p]) {
                if (cornerOf[i] < 0) return "Did not find any eligible corner points in edge mesh";
                for (int d = 0; d < 3; ++d) o.cornerPoints[3 * (size_t)p + d] = o.target.pts[3 * (size_t)cornerOf[i] + d];
                ++o.nCorner;
            }
            if (o.isFeatureEdgePoint[p]) ++o.nFeature;
        }
        if (o.enabled && metSmoothing[i]) { o.isSmoothingSurfacePoint[p] = 1; ++o.nSmoothingSurface; }
        else { o.isFrozenSurfacePoint[p] = 1; ++o.nFrozenSurface; }
    }

    // calculatePointHopsToBoundary(smoothingPatchIds, ..., maxIter = 2) OBB.C:52-133, SM.C:2218
    std::vector<int32_t>& hops = o.hopsToSmoothingBoundary;
    hops.assign((size_t)P, -1);
    for (const BndPatch& pp : patches) {
        if (!pp.isSmoothing) continue;
        for (int f = pp.start; f < pp.start + pp.size; ++f)
            for (int k = fp.off[f]; k < fp.off[f + 1]; ++k)
                if (o.isConnectedToInternalPoint[fp.val[k]]) hops[fp.val[k]] = 0;
    }
    o.hopsFresh.assign((size_t)P, -1);
    o.distanceTolerance = tol;
    return "";
}

// one sweep of calculatePointHopsToBoundary for the smoothing patches (OBB.C:85-121; under -parallel the host applies the
// maxEq sync :124-130 to the shared points between the sweeps)
void boundarySetupHopsSweep(const Topology& t, const uint8_t* internal, BoundarySetup& o) {
    const int P = t.nPoints;
    const Csr& pe = t.pointEdges;
    std::vector<int32_t>& hops = o.hopsToSmoothingBoundary;
    for (int p = 0; p < P; ++p) {
        if (hops[p] >= 0 || !internal[p]) continue;
        int mx = -1;
        for (int j = pe.off[p]; j < pe.off[p + 1]; ++j) mx = std::max(mx, hops[t.pointPoints[j]]);
        if (mx >= 0) o.hopsFresh[p] = mx + 1;
    }
    for (int p = 0; p < P; ++p) if (o.hopsFresh[p] > hops[p]) hops[p] = o.hopsFresh[p];
}

// propagateInnerNeighInfo OBB.C:396-459 and the target edge string of every feature edge point SM.C:2234-2249 (rank-local)
std::string boundarySetupFinish(const Topology& t, const double* points, BoundarySetup& o) {
    const int P = t.nPoints;
    const Csr& pe = t.pointEdges;
    const std::vector<int32_t>& hops = o.hopsToSmoothingBoundary;
    o.innerMap.assign((size_t)P, -1);
    for (int p = 0; p < P; ++p) {
        if (!o.isSmoothingSurfacePoint[p] || !o.isConnectedToInternalPoint[p]) continue;
        if (hops[p] != 0) return std::to_string(p) + " is not boundary point";
        int n = 0, q = -1;
        for (int j = pe.off[p]; j < pe.off[p + 1]; ++j)
            if (hops[t.pointPoints[j]] == 1) { ++n; q = t.pointPoints[j]; }
        if (n == 1) o.innerMap[p] = q;
    }
    if (o.enabled)
        for (int p = 0; p < P; ++p)
            if (o.isFeatureEdgePoint[p]) o.pointStrings[p] = closestEdge(ld(points, p), o.target, o.targetEdgeStrings, o.distanceTolerance).string;
    return "";
}

std::string buildBoundarySetupSerial(const Topology& t, const uint8_t* internal, const double* points, const std::vector<BndPatch>& patches,
                                     const BoundaryInputHost& in, BoundarySetup& o) {
    const std::string err = buildBoundarySetup(t, internal, points, patches, in, o);
    if (!err.empty()) return err;
    for (int iter = 0; iter < 2; ++iter) boundarySetupHopsSweep(t, internal, o);   // SM.C:2218
    return boundarySetupFinish(t, points, o);
}

// ---- bounding volume hierarchy over the target triangles -------------------------------------------------------------
void Bvh::build(const std::vector<double>& pts, const std::vector<int32_t>& tris) {
    const int n = (int)(tris.size() / 3);
    box.clear(); link.clear(); triVerts.clear(); triId.clear();
    if (n == 0) return;
    std::vector<double> lo(3 * (size_t)n), hi(3 * (size_t)n), ctr(3 * (size_t)n);
    double sceneLo[3] = {kVGreat, kVGreat, kVGreat}, sceneHi[3] = {-kVGreat, -kVGreat, -kVGreat};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const double a = pts[3 * (size_t)tris[3 * (size_t)i] + d], b = pts[3 * (size_t)tris[3 * (size_t)i + 1] + d], c = pts[3 * (size_t)tris[3 * (size_t)i + 2] + d];
            lo[3 * (size_t)i + d] = std::min(a, std::min(b, c));
            hi[3 * (size_t)i + d] = std::max(a, std::max(b, c));
            ctr[3 * (size_t)i + d] = (a + b + c) / 3.0;
            sceneLo[d] = std::min(sceneLo[d], lo[3 * (size_t)i + d]);
            sceneHi[d] = std::max(sceneHi[d], hi[3 * (size_t)i + d]);
        }
    // Hits are accepted a little outside a triangle (parametric tolerance 1e-14): the boxes are inflated by far more
    // than that, so the traversal can only test more triangles than needed, never fewer.
    double extent = 0.0;
    for (int d = 0; d < 3; ++d) extent = std::max(extent, std::max(std::abs(sceneLo[d]), std::abs(sceneHi[d])));
    const double pad = 1e-9 * std::max(extent, 1e-300);
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    struct Task { int node, first, count; };
    std::vector<Task> todo;
    box.resize(6); link.resize(2);
    todo.push_back({0, 0, n});
    const int leafSize = 4;
    while (!todo.empty()) {
        const Task tk = todo.back();
        todo.pop_back();
        double bl[3] = {kVGreat, kVGreat, kVGreat}, bh[3] = {-kVGreat, -kVGreat, -kVGreat};
        double cl[3] = {kVGreat, kVGreat, kVGreat}, ch[3] = {-kVGreat, -kVGreat, -kVGreat};
        for (int k = tk.first; k < tk.first + tk.count; ++k) {
            const int i = order[(size_t)k];
            for (int d = 0; d < 3; ++d) {
                bl[d] = std::min(bl[d], lo[3 * (size_t)i + d]); bh[d] = std::max(bh[d], hi[3 * (size_t)i + d]);
                cl[d] = std::min(cl[d], ctr[3 * (size_t)i + d]); ch[d] = std::max(ch[d], ctr[3 * (size_t)i + d]);
            }
        }
        for (int d = 0; d < 3; ++d) { box[6 * (size_t)tk.node + d] = bl[d] - pad; box[6 * (size_t)tk.node + 3 + d] = bh[d] + pad; }
        int axis = 0;
        for (int d = 1; d < 3; ++d) if (ch[d] - cl[d] > ch[axis] - cl[axis]) axis = d;
        if (tk.count <= leafSize) {   // leaves hold at most four triangles (the device loads a leaf in one batch)
            link[2 * (size_t)tk.node] = -(tk.first + 1);
            link[2 * (size_t)tk.node + 1] = tk.count;
            continue;
        }
        const int half = tk.count / 2;
        if (ch[axis] - cl[axis] > 0.0)   // coincident centroids: any split will do
            std::nth_element(order.begin() + tk.first, order.begin() + tk.first + half, order.begin() + tk.first + tk.count,
                         [&](int a, int b) { return ctr[3 * (size_t)a + axis] < ctr[3 * (size_t)b + axis] || (ctr[3 * (size_t)a + axis] == ctr[3 * (size_t)b + axis] && a < b); });
        const int left = (int)(link.size() / 2), right = left + 1;
        box.resize(box.size() + 12); link.resize(link.size() + 4);
        link[2 * (size_t)tk.node] = left;
        link[2 * (size_t)tk.node + 1] = right;
        todo.push_back({left, tk.first, half});
        todo.push_back({right, tk.first + half, tk.count - half});
    }
    // collapse into 8-wide nodes: the children of a wide node are the binary descendants reached by repeatedly opening
    // the internal candidate with the largest box surface until eight candidates stand (or only leaves are left)
    wideBox.clear(); wideRef.clear();
    wideDepth = 0;
    {
        auto area = [&](int node) {
            const double* b = &box[6 * (size_t)node];
            const double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            return dx * dy + dy * dz + dz * dx;
        };
        struct WTask { int binNode, wide, depth; };
        std::vector<WTask> wtodo;
        auto newWide = [&]() {
            const int w = (int)(wideRef.size() / 16);
            wideBox.resize(wideBox.size() + 48, 0.0f);
            wideRef.resize(wideRef.size() + 16, 0);
            for (int c = 0; c < 8; ++c) wideRef[16 * (size_t)w + 8 + (size_t)c] = -1;
            return w;
        };
        auto down = [](double v) { float f = (float)v; return ((double)f > v) ? std::nextafterf(f, -INFINITY) : f; };
        auto up = [](double v) { float f = (float)v; return ((double)f < v) ? std::nextafterf(f, INFINITY) : f; };
        wtodo.push_back({0, newWide(), 1});
        while (!wtodo.empty()) {
            const WTask tk = wtodo.back();
            wtodo.pop_back();
            wideDepth = std::max(wideDepth, tk.depth);
            std::vector<int> cand;
            if (link[2 * (size_t)tk.binNode] < 0) cand.push_back(tk.binNode);   // the whole (sub)tree is one leaf
            else { cand.push_back(link[2 * (size_t)tk.binNode]); cand.push_back(link[2 * (size_t)tk.binNode + 1]); }
            while (cand.size() < 8) {
                int best = -1;
                for (size_t c = 0; c < cand.size(); ++c)
                    if (link[2 * (size_t)cand[c]] >= 0 && (best < 0 || area(cand[c]) > area(cand[(size_t)best]))) best = (int)c;
                if (best < 0) break;
                const int open = cand[(size_t)best];
                cand[(size_t)best] = link[2 * (size_t)open];
                cand.push_back(link[2 * (size_t)open + 1]);
            }
            for (size_t c = 0; c < cand.size(); ++c) {
                const int bn = cand[c];
                for (int d = 0; d < 3; ++d) {
                    wideBox[48 * (size_t)tk.wide + 8 * (size_t)d + c] = down(box[6 * (size_t)bn + (size_t)d]);
                    wideBox[48 * (size_t)tk.wide + 24 + 8 * (size_t)d + c] = up(box[6 * (size_t)bn + 3 + (size_t)d]);
                }
                if (link[2 * (size_t)bn] < 0) {
                    wideRef[16 * (size_t)tk.wide + c] = -(link[2 * (size_t)bn] + 1);
                    wideRef[16 * (size_t)tk.wide + 8 + c] = link[2 * (size_t)bn + 1];
                } else {
                    const int w = newWide();
                    wideRef[16 * (size_t)tk.wide + c] = w;
                    wideRef[16 * (size_t)tk.wide + 8 + c] = 0;
                    wtodo.push_back({bn, w, tk.depth + 1});
                }
            }
        }
    }
    triVerts.assign(10 * (size_t)n, 0.0);
    triId.resize((size_t)n);
    for (int k = 0; k < n; ++k) {
        const int i = order[(size_t)k];
        triId[(size_t)k] = i;
        for (int v = 0; v < 3; ++v)
            for (int d = 0; d < 3; ++d) triVerts[10 * (size_t)k + 3 * (size_t)v + d] = pts[3 * (size_t)tris[3 * (size_t)i + v] + d];
        const int64_t idBits = i;
        std::memcpy(&triVerts[10 * (size_t)k + 9], &idBits, sizeof(double));
    }
}

}  // namespace smgpu
