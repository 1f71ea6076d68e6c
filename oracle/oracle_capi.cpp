// oracle_capi.cpp -- plain C entry points over the oracle for ctypes (tests, smoke, bench's
// cpu_baseline leg only).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see smooth_oracle.hpp).
#include <cstring>
#include <string>

#include <cmath>

#include "smooth_oracle.hpp"
#include "../smoothmesh_amd/csrc/smacos.hpp"

using namespace orc;

extern "C" {

void* orc_create(int nPoints, int nCells, int nFaces, int nInternalFaces, const double* points,
                 const int* faceOffsets, const int* facePoints, const int* owner, const int* neighbour,
                 const unsigned char* isInternalPoint, const unsigned char* isSmoothingSurfacePoint) {
    Domain* d = new Domain();
    d->nPoints = nPoints;
    d->nCells = nCells;
    d->points.resize(nPoints);
    std::memcpy(d->points.data(), points, sizeof(double) * 3 * size_t(nPoints));
    d->faces.resize(nFaces);
    for (int f = 0; f < nFaces; ++f) d->faces[f].assign(facePoints + faceOffsets[f], facePoints + faceOffsets[f + 1]);
    d->owner.assign(owner, owner + nFaces);
    d->neighbour.assign(neighbour, neighbour + nInternalFaces);
    d->isInternalPoint.assign(isInternalPoint, isInternalPoint + nPoints);
    if (isSmoothingSurfacePoint) d->isSmoothingSurfacePoint.assign(isSmoothingSurfacePoint, isSmoothingSurfacePoint + nPoints);
    else d->isSmoothingSurfacePoint.assign(nPoints, 0);
    d->build();
    return d;
}

void orc_destroy(void* h) { delete static_cast<Domain*>(h); }

// process-wide: 0 = glibc acos (the reference's arithmetic), 1 = the device kernels' algorithm (smacos.hpp)
void orc_set_acos_variant(int variant) { setAcosVariant(variant); }
int orc_get_acos_variant() { return acosVariant(); }
void orc_acos_census_enable(int on) { censusEnable(on != 0); if (on) censusReset(); }
// out = {comparisons, of them with equal sides, with sides 1..8 ulp apart, smallest distance of unequal sides in ulp (or -1)}
void orc_acos_census(long long* out) {
    const AcosCensus c = censusGet();
    out[0] = c.comparisons; out[1] = c.equal; out[2] = c.within8ulp;
    out[3] = (c.minUlp == ~0ull || c.minUlp > 0x7fffffffffffffffull) ? -1 : (long long)c.minUlp;
}
// the engine's near-tie census on the oracle's side: window in ulp (default 4), counts by class {SM.C:923, SM.C:1367 per point, walk verdicts}
void orc_acos_census_window(long long ulps) { censusWindow(ulps < 0 ? 0ull : (unsigned long long)ulps); }
void orc_acos_census_near(long long* out) { const AcosCensus c = censusGet(); out[0] = c.near[0]; out[1] = c.near[1]; out[2] = c.near[2]; }
double orc_acos(double x, int variant) { return variant ? smacos::acosX(x) : std::acos(x); }
void orc_set_foam_variant(void* h, int variant) { static_cast<Domain*>(h)->foamVariant = variant; }
// syncPointList model of the rank-engine combines below: 0 = master fold (globalMeshData::syncData), 1 = own-value fold
void orc_set_sync_variant(void* h, int variant) { static_cast<Domain*>(h)->syncVariant = variant; }
void orc_set_params(void* h, double maxStepLength, double relStepFrac, double minEdgeLength, int totalMinFreeze,
                    int edgeAngleConstraint, int faceAngleConstraint, double minAngle, double maxAngle) {
    Domain* d = static_cast<Domain*>(h);
    d->prm.maxStepLength = maxStepLength;
    d->prm.relStepFrac = relStepFrac;
    d->prm.minEdgeLength = minEdgeLength;
    d->prm.totalMinFreeze = totalMinFreeze != 0;
    d->prm.edgeAngleConstraint = edgeAngleConstraint != 0;
    d->prm.faceAngleConstraint = faceAngleConstraint != 0;
    d->prm.minAngle = minAngle;
    d->prm.maxAngle = maxAngle;
}

void orc_mesh_stats(void* h, double* minEdge, double* maxEdge) { static_cast<Domain*>(h)->meshStats(*minEdge, *maxEdge); }

int orc_iterate(void* h, int nIters, double relTol, double* residuals, int* nFrozen) {
    return static_cast<Domain*>(h)->iterate(nIters, relTol, residuals, nFrozen);
}

const char* orc_last_error(void* h) { return static_cast<Domain*>(h)->error.c_str(); }

void orc_set_points(void* h, const double* pts) {
    Domain* d = static_cast<Domain*>(h);
    std::memcpy(d->points.data(), pts, sizeof(double) * 3 * size_t(d->nPoints));
}

int orc_num_edges(void* h) { return int(static_cast<Domain*>(h)->edges.size()); }

// run one iteration's phases without committing (operator-level parity checks)
void orc_phaseA(void* h) { static_cast<Domain*>(h)->phaseA(); }
void orc_phaseB(void* h) { static_cast<Domain*>(h)->phaseB(); }
void orc_phaseC(void* h) { static_cast<Domain*>(h)->phaseC(); }
void orc_commit(void* h) { static_cast<Domain*>(h)->commit(); }

static long long copyVec(const std::vector<Vec3>& v, double* out) {
    if (out) std::memcpy(out, v.data(), sizeof(Vec3) * v.size());
    return (long long)v.size() * 3;
}
static long long copyD(const std::vector<double>& v, double* out) {
    if (out) std::memcpy(out, v.data(), sizeof(double) * v.size());
    return (long long)v.size();
}
static long long copyU8(const std::vector<unsigned char>& v, double* out) {
    if (out) for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (long long)v.size();
}
static long long copyI(const std::vector<int>& v, double* out) {
    if (out) for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (long long)v.size();
}

// Copy a named field as doubles; returns its length (call with out == NULL to size it), -1 if unknown.
long long orc_get_field(void* h, const char* name, double* out) {
    Domain* d = static_cast<Domain*>(h);
    const std::string n(name);
    if (n == "points") return copyVec(d->points, out);
    if (n == "faceCentres") return copyVec(d->faceCentres, out);
    if (n == "faceAreas") return copyVec(d->faceAreas, out);
    if (n == "cellCentres") return copyVec(d->cellCentres, out);
    if (n == "cellSum") return copyVec(d->cellSum, out);
    if (n == "cellCount") return copyI(d->cellCount, out);
    if (n == "closest1") return copyVec(d->closest1, out);
    if (n == "closest2") return copyVec(d->closest2, out);
    if (n == "closest3") return copyVec(d->closest3, out);
    if (n == "hasCommonCell") return copyU8(d->hasCommonCell, out);
    if (n == "centroidalPoints") return copyVec(d->centroidalPoints, out);
    if (n == "arPoints") return copyVec(d->arPoints, out);
    if (n == "newPoints") return copyVec(d->newPoints, out);
    if (n == "isFrozenPoint") return copyU8(d->isFrozenPoint, out);
    if (n == "frozenAfterEdgeLen") return copyU8(d->frozenAfterEdgeLen, out);
    if (n == "frozenAfterEdgeAngle") return copyU8(d->frozenAfterEdgeAngle, out);
    if (n == "frozenAfterFaceAngle") return copyU8(d->frozenAfterFaceAngle, out);
    if (n == "edgeMinAngle") return copyD(d->edgeMinAngle, out);
    if (n == "edgeMaxAngle") return copyD(d->edgeMaxAngle, out);
    if (n == "pointMinAngle") return copyD(d->pointMinAngle, out);
    if (n == "pointMaxAngle") return copyD(d->pointMaxAngle, out);
    if (n == "eaMinC") return copyD(d->eaMinC, out);
    if (n == "eaMinN") return copyD(d->eaMinN, out);
    return -1;
}

// Addressing as CSR (offsets has n+1 entries).  kind: pointCells, pointFaces, pointEdges, pointPoints,
// edgeFaces, edgeCells, cellFaces, edges (returned as 2 per row).  Returns nnz; call with NULLs to size.
long long orc_get_addressing(void* h, const char* kind, int* offsets, int* values) {
    Domain* d = static_cast<Domain*>(h);
    const std::string n(kind);
    const std::vector<std::vector<int>>* ll = nullptr;
    if (n == "pointCells") ll = &d->pointCells;
    else if (n == "pointFaces") ll = &d->pointFaces;
    else if (n == "pointEdges") ll = &d->pointEdges;
    else if (n == "pointPoints") ll = &d->pointPoints;
    else if (n == "edgeFaces") ll = &d->edgeFaces;
    else if (n == "edgeCells") ll = &d->edgeCells;
    else if (n == "cellFaces") ll = &d->cellFaces;
    else if (n == "edges") {
        if (values) for (size_t e = 0; e < d->edges.size(); ++e) { values[2 * e] = d->edges[e][0]; values[2 * e + 1] = d->edges[e][1]; }
        return (long long)d->edges.size() * 2;
    } else return -1;
    long long nnz = 0;
    for (size_t i = 0; i < ll->size(); ++i) {
        if (offsets) offsets[i] = int(nnz);
        for (int v : (*ll)[i]) { if (values) values[nnz] = v; ++nnz; }
    }
    if (offsets) offsets[ll->size()] = int(nnz);
    return nnz;
}

// scalar building blocks for known-answer tests
double orc_edgeEdgeAngle(const double* c, const double* p1, const double* p2) {
    return edgeEdgeAngle({c[0], c[1], c[2]}, {p1[0], p1[1], p1[2]}, {p2[0], p2[1], p2[2]});
}
double orc_calcEdgeCenterEdgeAngle(const double* p0, const double* cC, const double* p1) {
    return calcEdgeCenterEdgeAngle({p0[0], p0[1], p0[2]}, {cC[0], cC[1], cC[2]}, {p1[0], p1[1], p1[2]});
}
int orc_isCloserPoint(const double* a, const double* b) {
    return isCloserPoint({a[0], a[1], a[2]}, {b[0], b[1], b[2]}) ? 1 : 0;
}

// ---- multi-domain ----------------------------------------------------------------------
void* orc_multi_create(int nDomains, void** handles) {
    MultiDomain* m = new MultiDomain();
    for (int i = 0; i < nDomains; ++i) m->dom.push_back(static_cast<Domain*>(handles[i]));
    return m;
}
void orc_multi_destroy(void* m) { delete static_cast<MultiDomain*>(m); }
void orc_multi_set_sync_variant(void* m, int variant) { static_cast<MultiDomain*>(m)->syncVariant = variant; }
// shared points as CSR: for shared point s, entries [off[s], off[s+1]) give (domain, local) pairs,
// ascending domain id
void orc_multi_set_shared(void* mh, int nShared, const int* off, const int* domain, const int* local) {
    MultiDomain* m = static_cast<MultiDomain*>(mh);
    m->shared.resize(nShared);
    for (int s = 0; s < nShared; ++s) {
        m->shared[s].domain.assign(domain + off[s], domain + off[s + 1]);
        m->shared[s].local.assign(local + off[s], local + off[s + 1]);
    }
}
int orc_multi_iterate(void* mh, int nIters, double relTol, double* residuals, int* nFrozen) {
    return static_cast<MultiDomain*>(mh)->iterate(nIters, relTol, residuals, nFrozen);
}

// ---- one rank's side of the shared-point exchange (used by the CPU multi-process tests, where an
// oracle Domain stands in for the HIP engine behind smoothmesh_amd.halo) -----------------------
// record layout = include/smgpu.h exchange A: 13 doubles {sum, r1, r2, r3, (int32 count | int32 hc << 32)}
static void packRecord(const Domain* d, int p, double* r) {
    const Vec3 &s = d->cellSum[p], &a = d->closest1[p], &b = d->closest2[p], &c = d->closest3[p];
    r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = a.x; r[4] = a.y; r[5] = a.z;
    r[6] = b.x; r[7] = b.y; r[8] = b.z; r[9] = c.x; r[10] = c.y; r[11] = c.z;
    const long long pk = ((long long)d->hasCommonCell[p] << 32) | (long long)(unsigned int)d->cellCount[p];
    std::memcpy(&r[12], &pk, 8);
}
void orc_halo_packA(void* h, int nShared, const int* sharedLocal, int nSend, const int* sendShared, double* sendA) {
    Domain* d = static_cast<Domain*>(h);
    (void)nShared;
    for (int i = 0; i < nSend; ++i) packRecord(d, sharedLocal[sendShared[i]], sendA + 13 * (size_t)i);
}
void orc_halo_combineA(void* h, int nShared, const int* sharedLocal, const int* combOff, const int* combSlots, const double* recvA) {
    Domain* d = static_cast<Domain*>(h);
    for (int i = 0; i < nShared; ++i) {
        const int p = sharedLocal[i];
        const int b = combOff[i], n = combOff[i + 1] - b;
        std::vector<Vec3> r1(n), r2(n), r3(n);
        std::vector<unsigned char> hc(n);
        Vec3 sum{0, 0, 0};
        int cnt = 0, self = 0;
        double own[13];
        packRecord(d, p, own);
        for (int j = 0; j < n; ++j) {
            const int sl = combSlots[b + j];
            const double* r = (sl < 0) ? own : recvA + 13 * (size_t)sl;
            if (sl < 0) self = j;
            sum.x += r[0]; sum.y += r[1]; sum.z += r[2];
            r1[j] = {r[3], r[4], r[5]}; r2[j] = {r[6], r[7], r[8]}; r3[j] = {r[9], r[10], r[11]};
            long long pk; std::memcpy(&pk, &r[12], 8);
            cnt += (int)(pk & 0xffffffffll);
            hc[j] = (unsigned char)(pk >> 32);
        }
        combineClosest(n, r1.data(), r2.data(), r3.data(), hc.data(), d->syncVariant);
        unsigned char any = 0;
        for (int j = 0; j < n; ++j) any |= hc[j];
        d->cellSum[p] = sum; d->cellCount[p] = cnt;
        d->closest1[p] = r1[self]; d->closest2[p] = r2[self]; d->closest3[p] = r3[self];
        d->hasCommonCell[p] = any;
    }
}
// boundary layer treatment under -parallel, rank-engine form: 6 doubles per shared point (local normal, local outer
// neighbour coordinates), combined as MultiDomain::syncLayers does
// the exchange-L record of include/smgpu.h (SMGPU_HALO_L_DOUBLES = 14): normal, outer neighbour coordinates, local face
// count, inner neighbour coordinates, local feature edge projection sum and count
static const int kL = 14;
void orc_halo_packL(void* h, const int* sharedLocal, int nSend, const int* sendShared, double* sendL) {
    Domain* d = static_cast<Domain*>(h);
    const bool bnd = d->doBoundarySmoothing;
    for (int i = 0; i < nSend; ++i) {
        const int p = sharedLocal[sendShared[i]];
        double* r = sendL + (size_t)kL * (size_t)i;
        r[0] = d->pointNormals[p].x; r[1] = d->pointNormals[p].y; r[2] = d->pointNormals[p].z;
        r[3] = d->outerNeighCoords[p].x; r[4] = d->outerNeighCoords[p].y; r[5] = d->outerNeighCoords[p].z;
        r[6] = d->layerNFaces.empty() ? 0.0 : (double)d->layerNFaces[p];
        r[7] = r[8] = r[9] = GREAT;
        r[10] = r[11] = r[12] = r[13] = 0.0;
        if (bnd) {
            r[7] = d->innerNeighCoords[p].x; r[8] = d->innerNeighCoords[p].y; r[9] = d->innerNeighCoords[p].z;
            r[10] = d->featureEdgeProjections[p].x; r[11] = d->featureEdgeProjections[p].y; r[12] = d->featureEdgeProjections[p].z;
            r[13] = (double)d->nFeatureEdgeProjections[p];
        }
    }
}
void orc_halo_combineL(void* h, int nShared, const int* sharedLocal, const int* combOff, const int* combSlots, const double* recvL) {
    Domain* d = static_cast<Domain*>(h);
    const bool bnd = d->doBoundarySmoothing;
    auto fold = [](const Vec3& x, const Vec3& y) {
        const double mx = x.x * x.x + x.y * x.y + x.z * x.z, my = y.x * y.x + y.y * y.y + y.z * y.z;
        return (mx <= my) ? x : y;
    };
    for (int i = 0; i < nShared; ++i) {
        const int p = sharedLocal[i];
        const int b = combOff[i], n = combOff[i + 1] - b;
        const Vec3 ownN = d->pointNormals[p];
        Vec3 sum{0, 0, 0}, fsum{0, 0, 0};
        // minMagSqrEqOp: master fold (the first sharer's value, the others folded onto it in ascending rank order; every sharer
        // gets the same result) or, syncVariant 1, folded onto the own value
        const bool ownFold = d->syncVariant == 1;
        const Vec3 ownX = d->outerNeighCoords[p], ownY = bnd ? d->innerNeighCoords[p] : Vec3{GREAT, GREAT, GREAT};
        Vec3 x = ownX, y = ownY;
        int faces = 0, fcnt = 0;
        for (int j = 0; j < n; ++j) {
            const int sl = combSlots[b + j];
            const double* r = (sl < 0) ? nullptr : recvL + (size_t)kL * (size_t)sl;
            const Vec3 nj = r ? Vec3{r[0], r[1], r[2]} : ownN;
            sum.x += nj.x; sum.y += nj.y; sum.z += nj.z;
            faces += r ? (int)r[6] : (d->layerNFaces.empty() ? 0 : d->layerNFaces[p]);
            if (bnd) {
                const Vec3 fj = r ? Vec3{r[10], r[11], r[12]} : d->featureEdgeProjections[p];
                fsum.x += fj.x; fsum.y += fj.y; fsum.z += fj.z;
                fcnt += r ? (int)r[13] : d->nFeatureEdgeProjections[p];
            }
            const Vec3 xj = r ? Vec3{r[3], r[4], r[5]} : ownX;
            const Vec3 yj = r ? (bnd ? Vec3{r[7], r[8], r[9]} : Vec3{GREAT, GREAT, GREAT}) : ownY;
            if (ownFold) {
                if (r) { x = fold(x, xj); if (bnd) y = fold(y, yj); }
            } else if (j == 0) { x = xj; y = yj; }
            else { x = fold(x, xj); if (bnd) y = fold(y, yj); }
        }
        d->pointNormals[p] = sum;
        d->outerNeighCoords[p] = x;
        if (!d->layerNFaces.empty()) d->layerNFaces[p] = faces;
        if (bnd) { d->innerNeighCoords[p] = y; d->featureEdgeProjections[p] = fsum; d->nFeatureEdgeProjections[p] = fcnt; }
    }
}
// step-wise set-up (same steps / fields as include/smgpu.h smgpu_layers_*)
int orc_layers_begin(void* h, int nPatches, const int* start, const int* size, const int* kind, const unsigned char* isLayer,
                     double layerMaxBlendingFraction, double layerEdgeLength, double layerExpansionRatio, int minLayers, int maxLayers) {
    Domain* d = static_cast<Domain*>(h);
    std::vector<Patch> p((size_t)nPatches);
    for (int i = 0; i < nPatches; ++i) { p[i].start = start[i]; p[i].size = size[i]; p[i].kind = kind[i]; p[i].isLayerPatch = isLayer[i] != 0; }
    LayerParams lp;
    lp.layerMaxBlendingFraction = layerMaxBlendingFraction;
    lp.layerEdgeLength = layerEdgeLength;
    lp.layerExpansionRatio = layerExpansionRatio;
    lp.minLayers = minLayers;
    lp.maxLayers = maxLayers;
    d->layersBegin(p, lp);
    return d->doLayerTreatment ? 1 : 0;
}
void orc_layers_step(void* h, int step, int arg) {
    Domain* d = static_cast<Domain*>(h);
    switch (step) {
    case 0: d->layersHopsSweep(); break;
    case 1: d->layersNormalsAccumulate(); break;
    case 2: d->layersNormalsFinish(); break;
    case 3: d->layersPropagateSweep(arg); break;
    case 4: d->layersUndo(); break;
    }
}
void orc_layers_shared(void* h, int nShared, const int* sharedLocal, int field, int set, double* v) {
    Domain* d = static_cast<Domain*>(h);
    for (int i = 0; i < nShared; ++i) {
        const int p = sharedLocal[i];
        if (field == 0) {
            if (set) d->pointHopsToLayerBoundary[p] = (int)v[i]; else v[i] = d->pointHopsToLayerBoundary[p];
        } else {
            const int w = field == 1 ? 4 : 3;
            Vec3& n = d->pointNormals[p];
            if (set) { n.x = v[w * i]; n.y = v[w * i + 1]; n.z = v[w * i + 2]; } else { v[w * i] = n.x; v[w * i + 1] = n.y; v[w * i + 2] = n.z; }
            if (w == 4) { if (set) d->layerNFaces[p] = (int)v[4 * i + 3]; else v[4 * i + 3] = d->layerNFaces[p]; }
        }
    }
}

void orc_halo_packF(void* h, const int* sharedLocal, int nSend, const int* sendShared, int* sendF) {
    Domain* d = static_cast<Domain*>(h);
    for (int i = 0; i < nSend; ++i) sendF[i] = d->isFrozenPoint[sharedLocal[sendShared[i]]];
}
void orc_halo_orF(void* h, int nShared, const int* sharedLocal, const int* combOff, const int* combSlots, const int* recvF) {
    Domain* d = static_cast<Domain*>(h);
    for (int i = 0; i < nShared; ++i)
        for (int k = combOff[i]; k < combOff[i + 1]; ++k)
            if (combSlots[k] >= 0 && recvF[combSlots[k]]) d->isFrozenPoint[sharedLocal[i]] = 1;
}
void orc_local_stats(void* h, double* out2) {
    Domain* d = static_cast<Domain*>(h);
    out2[0] = d->residualLocal; out2[1] = (double)d->nFrozenLocal;
}

// boundary layer treatment (serial): patches as (start, size, kind 0/1/2, isLayerPatch)
void orc_setup_layers(void* h, int nPatches, const int* start, const int* size, const int* kind, const unsigned char* isLayer,
                      double layerMaxBlendingFraction, double layerEdgeLength, double layerExpansionRatio, int minLayers,
                      int maxLayers) {
    Domain* d = static_cast<Domain*>(h);
    std::vector<Patch> p((size_t)nPatches);
    for (int i = 0; i < nPatches; ++i) { p[i].start = start[i]; p[i].size = size[i]; p[i].kind = kind[i]; p[i].isLayerPatch = isLayer[i] != 0; }
    LayerParams lp;
    lp.layerMaxBlendingFraction = layerMaxBlendingFraction;
    lp.layerEdgeLength = layerEdgeLength;
    lp.layerExpansionRatio = layerExpansionRatio;
    lp.minLayers = minLayers;
    lp.maxLayers = maxLayers;
    d->setupLayers(p, lp);
}
int orc_layers_enabled(void* h) { return static_cast<Domain*>(h)->doLayerTreatment ? 1 : 0; }
void orc_get_layer_fields(void* h, int* hops, int* outerMap, double* normals, unsigned char* isConnected, unsigned char* isLayerSurface) {
    Domain* d = static_cast<Domain*>(h);
    for (int p = 0; p < d->nPoints; ++p) {
        hops[p] = d->pointHopsToLayerBoundary[p];
        outerMap[p] = d->pointToOuterPointMap[p];
        normals[3 * p] = d->pointNormals[p].x; normals[3 * p + 1] = d->pointNormals[p].y; normals[3 * p + 2] = d->pointNormals[p].z;
        isConnected[p] = d->isConnectedToInternalPoint[p];
        isLayerSurface[p] = d->isLayerSurfacePoint[p];
    }
}

// boundary point smoothing (serial), SM.C:2080-2253: patches + layer options as orc_setup_layers, edge meshes and target
// surface as flat arrays (targetEdges may be empty: the initial edges are then the target, SM.C:2154-2160), the
// classification lists of a previous run or NULL.  Returns doBoundarySmoothing, -1 on the reference's FatalErrors.
int orc_setup_boundary(void* h, int nPatches, const int* start, const int* size, const int* kind, const unsigned char* isLayer,
                       const unsigned char* isSmoothing, double layerMaxBlendingFraction, double layerEdgeLength,
                       double layerExpansionRatio, int minLayers, int maxLayers, int nInitPts, const double* initPts, int nInitEdges,
                       const int* initEdges, int nTgtPts, const double* tgtPts, int nTgtEdges, const int* tgtEdges, int nSurfPts,
                       const double* surfPts, int nTris, const int* tris, const int* cornerIO, const int* featureIO,
                       double internalSmoothingBlendingFraction) {
    Domain* d = static_cast<Domain*>(h);
    std::vector<Patch> p((size_t)nPatches);
    for (int i = 0; i < nPatches; ++i) {
        p[i].start = start[i]; p[i].size = size[i]; p[i].kind = kind[i];
        p[i].isLayerPatch = isLayer[i] != 0; p[i].isSmoothingPatch = isSmoothing[i] != 0;
    }
    LayerParams lp;
    lp.layerMaxBlendingFraction = layerMaxBlendingFraction;
    lp.layerEdgeLength = layerEdgeLength;
    lp.layerExpansionRatio = layerExpansionRatio;
    lp.minLayers = minLayers;
    lp.maxLayers = maxLayers;
    BoundaryInput in;
    auto fillEdges = [](EdgeMesh& em, int nP, const double* pts, int nE, const int* e) {
        em.points.resize((size_t)nP);
        for (int i = 0; i < nP; ++i) em.points[(size_t)i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        em.edges.resize((size_t)nE);
        for (int i = 0; i < nE; ++i) em.edges[(size_t)i] = {e[2 * i], e[2 * i + 1]};
    };
    fillEdges(in.initEdges, nInitPts, initPts, nInitEdges, initEdges);
    fillEdges(in.targetEdges, nTgtPts, tgtPts, nTgtEdges, tgtEdges);
    in.surf.points.resize((size_t)nSurfPts);
    for (int i = 0; i < nSurfPts; ++i) in.surf.points[(size_t)i] = {surfPts[3 * i], surfPts[3 * i + 1], surfPts[3 * i + 2]};
    in.surf.tris.resize((size_t)nTris);
    for (int i = 0; i < nTris; ++i) in.surf.tris[(size_t)i] = {tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]};
    if (cornerIO) in.isCornerPointIO.assign(cornerIO, cornerIO + d->nPoints);
    if (featureIO) in.isFeatureEdgePointIO.assign(featureIO, featureIO + d->nPoints);
    in.internalSmoothingBlendingFraction = internalSmoothingBlendingFraction;
    d->setupBoundary(p, lp, in);
    if (!d->error.empty()) return -1;
    return d->doBoundarySmoothing ? 1 : 0;
}
void orc_get_boundary_fields(void* h, unsigned char* isCorner, unsigned char* isFeature, unsigned char* isSmoothingSurface,
                             unsigned char* isSharp, double* cornerPoints, int* pointStrings, int* innerMap, int* hopsSmoothing,
                             double* normals) {
    Domain* d = static_cast<Domain*>(h);
    for (int p = 0; p < d->nPoints; ++p) {
        isCorner[p] = d->isCornerPoint[p]; isFeature[p] = d->isFeatureEdgePoint[p];
        isSmoothingSurface[p] = d->isSmoothingSurfacePoint[p]; isSharp[p] = d->isSharpEdgePoint[p];
        cornerPoints[3 * p] = d->cornerPoints[p].x; cornerPoints[3 * p + 1] = d->cornerPoints[p].y; cornerPoints[3 * p + 2] = d->cornerPoints[p].z;
        pointStrings[p] = d->pointStrings[p]; innerMap[p] = d->pointToInnerPointMap[p]; hopsSmoothing[p] = d->pointHopsToSmoothingBoundary[p];
        normals[3 * p] = d->pointNormals[p].x; normals[3 * p + 1] = d->pointNormals[p].y; normals[3 * p + 2] = d->pointNormals[p].z;
    }
}
int orc_get_edge_strings(void* h, int* out) {
    Domain* d = static_cast<Domain*>(h);
    if (out) for (size_t i = 0; i < d->targetEdgeStrings.size(); ++i) out[i] = d->targetEdgeStrings[i];
    return (int)d->targetEdgeStrings.size();
}
// findEdgeMeshStrings BPS.C:557-587 on a bare edge mesh; returns the number of strings
int orc_edge_strings(int nPts, int nEdges, const int* edges, int* out) {
    EdgeMesh em;
    em.points.assign((size_t)nPts, Vec3{0, 0, 0});
    em.edges.resize((size_t)nEdges);
    for (int i = 0; i < nEdges; ++i) em.edges[(size_t)i] = {edges[2 * i], edges[2 * i + 1]};
    em.buildPointEdges();
    std::vector<int> strings;
    const int last = findEdgeMeshStrings(strings, em);
    for (int i = 0; i < nEdges; ++i) out[i] = strings[(size_t)i];
    return last + 1;
}
// nearest hit of a segment with the target surface (Domain::findLine); returns 1 on a hit
int orc_find_line(void* h, const double* start, const double* end, double* hitPoint) {
    Domain* d = static_cast<Domain*>(h);
    bool hit;
    const Vec3 r = d->findLine({start[0], start[1], start[2]}, {end[0], end[1], end[2]}, hit);
    hitPoint[0] = r.x; hitPoint[1] = r.y; hitPoint[2] = r.z;
    return hit ? 1 : 0;
}

// MultiDomain variant: patches of all domains concatenated, nPatches[d] per domain
void orc_multi_setup_layers(void* mh, const int* nPatches, const int* start, const int* size, const int* kind, const unsigned char* isLayer,
                            double layerMaxBlendingFraction, double layerEdgeLength, double layerExpansionRatio, int minLayers,
                            int maxLayers) {
    MultiDomain* m = static_cast<MultiDomain*>(mh);
    std::vector<std::vector<Patch>> p(m->dom.size());
    int k = 0;
    for (size_t d = 0; d < m->dom.size(); ++d)
        for (int i = 0; i < nPatches[d]; ++i, ++k) {
            Patch q; q.start = start[k]; q.size = size[k]; q.kind = kind[k]; q.isLayerPatch = isLayer[k] != 0;
            p[d].push_back(q);
        }
    LayerParams lp;
    lp.layerMaxBlendingFraction = layerMaxBlendingFraction;
    lp.layerEdgeLength = layerEdgeLength;
    lp.layerExpansionRatio = layerExpansionRatio;
    lp.minLayers = minLayers;
    lp.maxLayers = maxLayers;
    m->setupLayers(p, lp);
}

// boundary point smoothing of a MultiDomain (-parallel): patches of all domains concatenated, nPatches[d] per domain; the
// geometry inputs are the same for every domain (each rank reads the same constant/geometry files).  Returns
// doBoundarySmoothing, -1 on the reference's FatalErrors.
int orc_multi_setup_boundary(void* mh, const int* nPatches, const int* start, const int* size, const int* kind, const unsigned char* isLayer,
                             const unsigned char* isSmoothing, double layerMaxBlendingFraction, double layerEdgeLength,
                             double layerExpansionRatio, int minLayers, int maxLayers, int nInitPts, const double* initPts, int nInitEdges,
                             const int* initEdges, int nTgtPts, const double* tgtPts, int nTgtEdges, const int* tgtEdges, int nSurfPts,
                             const double* surfPts, int nTris, const int* tris, double internalSmoothingBlendingFraction) {
    MultiDomain* m = static_cast<MultiDomain*>(mh);
    std::vector<std::vector<Patch>> p(m->dom.size());
    int k = 0;
    for (size_t d = 0; d < m->dom.size(); ++d)
        for (int i = 0; i < nPatches[d]; ++i, ++k) {
            Patch q; q.start = start[k]; q.size = size[k]; q.kind = kind[k]; q.isLayerPatch = isLayer[k] != 0; q.isSmoothingPatch = isSmoothing[k] != 0;
            p[d].push_back(q);
        }
    LayerParams lp;
    lp.layerMaxBlendingFraction = layerMaxBlendingFraction;
    lp.layerEdgeLength = layerEdgeLength;
    lp.layerExpansionRatio = layerExpansionRatio;
    lp.minLayers = minLayers;
    lp.maxLayers = maxLayers;
    BoundaryInput in;
    auto fillEdges = [](EdgeMesh& em, int nP, const double* pts, int nE, const int* e) {
        em.points.resize((size_t)nP);
        for (int i = 0; i < nP; ++i) em.points[(size_t)i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        em.edges.resize((size_t)nE);
        for (int i = 0; i < nE; ++i) em.edges[(size_t)i] = {e[2 * i], e[2 * i + 1]};
    };
    fillEdges(in.initEdges, nInitPts, initPts, nInitEdges, initEdges);
    fillEdges(in.targetEdges, nTgtPts, tgtPts, nTgtEdges, tgtEdges);
    in.surf.points.resize((size_t)nSurfPts);
    for (int i = 0; i < nSurfPts; ++i) in.surf.points[(size_t)i] = {surfPts[3 * i], surfPts[3 * i + 1], surfPts[3 * i + 2]};
    in.surf.tris.resize((size_t)nTris);
    for (int i = 0; i < nTris; ++i) in.surf.tris[(size_t)i] = {tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]};
    in.internalSmoothingBlendingFraction = internalSmoothingBlendingFraction;
    m->setupBoundary(p, lp, in);
    for (Domain* d : m->dom) if (!d->error.empty()) return -1;
    return (!m->dom.empty() && m->dom[0]->doBoundarySmoothing) ? 1 : 0;
}

}  // extern "C"

