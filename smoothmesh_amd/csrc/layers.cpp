// layers.cpp -- see layers.hpp.  Plain IEEE f64 in the reference's evaluation order (-ffp-contract=off).
#include "layers.hpp"

#include <algorithm>
#include <cmath>

namespace smgpu {
namespace {
const double kGreat = 1.0e15;   // UNDEF_VECTOR = (GREAT, GREAT, GREAT), COM.H:15

struct N3 { double x, y, z; };
inline N3 ld(const std::vector<double>& v, int p) { return {v[3 * (size_t)p], v[3 * (size_t)p + 1], v[3 * (size_t)p + 2]}; }
inline void st(std::vector<double>& v, int p, const N3& a) { v[3 * (size_t)p] = a.x; v[3 * (size_t)p + 1] = a.y; v[3 * (size_t)p + 2] = a.z; }
inline double magOf(const N3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline bool isZero(const N3& a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
inline bool isUndef(const N3& a) { return a.x == kGreat && a.y == kGreat && a.z == kGreat; }
}  // namespace

std::string LayerBuilder::begin(const Topology& t, const uint8_t* internal, const std::vector<LayerPatch>& patches, const double* faceArea,
                                double maxBlend, double edgeLength, double ratio, int minLayers, int maxLayers) {
    const int P = t.nPoints;
    for (const LayerPatch& pp : patches)
        if (pp.start < t.nInternalFaces || pp.size < 0 || pp.start + pp.size > t.nFaces) return "layer set-up: patch face range outside the boundary faces";
    if (maxLayers < 0 || minLayers < 0) return "layer set-up: minLayers / maxLayers must not be negative";
    t_ = &t; internal_ = internal; patches_ = patches; faceArea_ = faceArea;
    maxBlend_ = maxBlend; edgeLength_ = edgeLength; ratio_ = ratio; minLayers_ = minLayers; maxLayers_ = maxLayers;
    maxIter = maxLayers + 1;
    LayerSetup& o = out;
    o.hops.assign((size_t)P, -1);
    o.outerMap.assign((size_t)P, -1);
    o.normals.assign(3 * (size_t)P, 0.0);
    o.isConnectedToInternalPoint.assign((size_t)P, 0);
    o.isLayerSurfacePoint.assign((size_t)P, 0);
    nFaces.assign((size_t)P, 0);
    fresh_.assign((size_t)P, -1);
    // findIndex(boundaryPointLabels, n): claims of one neighbour come in one sweep in ascending order, so first = lowest
    firstClaim_.assign((size_t)P, -1);
    const Csr& fp = t.facePoints;
    const Csr& pe = t.pointEdges;   // pointPoints shares its offsets

    // BPS.C:296-340, 397-403: a boundary point is classified on the first patch that holds it
    std::vector<uint8_t> visited((size_t)P, 0);
    for (const LayerPatch& pp : patches)
        for (int f = pp.start; f < pp.start + pp.size; ++f)
            for (int k = fp.off[f]; k < fp.off[f + 1]; ++k) {
                const int p = fp.val[k];
                if (visited[p]) continue;
                visited[p] = 1;
                if (internal[p]) continue;
                for (int j = pe.off[p]; j < pe.off[p + 1]; ++j)
                    if (internal[t.pointPoints[j]]) { o.isConnectedToInternalPoint[p] = 1; break; }
                if (pp.isLayer) o.isLayerSurfacePoint[p] = 1;
            }
    // OBB.C:62-79: zero hops on the layer patches
    for (const LayerPatch& pp : patches) {
        if (!pp.isLayer) continue;
        for (int f = pp.start; f < pp.start + pp.size; ++f)
            for (int k = fp.off[f]; k < fp.off[f + 1]; ++k)
                if (o.isConnectedToInternalPoint[fp.val[k]]) o.hops[fp.val[k]] = 0;
    }
    return "";
}

// OBB.C:85-121: one sweep of the edge hop count.  Sweep k labels exactly the unlabelled internal points that touch a
// labelled point.
void LayerBuilder::hopsSweep() {
    const Topology& t = *t_;
    const Csr& pe = t.pointEdges;
    LayerSetup& o = out;
    for (int p = 0; p < t.nPoints; ++p) {
        if (o.hops[p] >= 0 || !internal_[p]) continue;
        int mx = -1;
        for (int j = pe.off[p]; j < pe.off[p + 1]; ++j) mx = std::max(mx, o.hops[t.pointPoints[j]]);
        if (mx >= 0) fresh_[p] = mx + 1;
    }
    for (int p = 0; p < t.nPoints; ++p)
        if (fresh_[p] > o.hops[p]) o.hops[p] = fresh_[p];
}

// OBB.C:150-181 on the current normals (zero at set-up, SM.C:1987): minus the sum of the unit normals of the point's
// boundary faces, in patch / face / vertex order
void LayerBuilder::normalsAccumulate() {
    const Topology& t = *t_;
    const Csr& fp = t.facePoints;
    LayerSetup& o = out;
    std::fill(nFaces.begin(), nFaces.end(), 0);
    for (const LayerPatch& pp : patches_) {
        if (pp.kind != 0) continue;
        for (int f = pp.start; f < pp.start + pp.size; ++f) {
            const N3 sf = {faceArea_[3 * (size_t)f], faceArea_[3 * (size_t)f + 1], faceArea_[3 * (size_t)f + 2]};
            const double m = magOf(sf);
            const N3 u = {sf.x / m, sf.y / m, sf.z / m};
            for (int k = fp.off[f]; k < fp.off[f + 1]; ++k) {
                const int p = fp.val[k];
                N3 n = ld(o.normals, p);
                n.x -= u.x; n.y -= u.y; n.z -= u.z;
                st(o.normals, p, n);
                ++nFaces[p];
            }
        }
    }
}

// OBB.C:201-230: shorter than 0.1 = sharp edge -> no normal; then unit length
void LayerBuilder::normalsFinish() {
    LayerSetup& o = out;
    const int P = t_->nPoints;
    for (int p = 0; p < P; ++p) {
        if (nFaces[p] < 1) continue;
        if (magOf(ld(o.normals, p)) < 0.1) st(o.normals, p, N3{0.0, 0.0, 0.0});
    }
    for (int p = 0; p < P; ++p) {
        const N3 n = ld(o.normals, p);
        if (isZero(n)) continue;
        const double m = magOf(n);
        st(o.normals, p, N3{n.x / m, n.y / m, n.z / m});
    }
}

// OBB.C:276-353: a point with exactly one neighbour one hop nearer to the patch hangs on that neighbour (a prismatic
// edge) and inherits its normal; a neighbour claimed twice disqualifies both claimants (and everything that later
// inherits from them, through the UNDEF marker)
void LayerBuilder::propagateSweep(int iter) {
    const Topology& t = *t_;
    const Csr& pe = t.pointEdges;
    LayerSetup& o = out;
    for (int p = 0; p < t.nPoints; ++p) {
        if (o.hops[p] != iter) continue;
        int cnt = 0, nb = -1;
        for (int j = pe.off[p]; j < pe.off[p + 1]; ++j)
            if (o.hops[t.pointPoints[j]] == iter - 1) { ++cnt; nb = t.pointPoints[j]; }
        if (cnt != 1) continue;
        if (!internal_[nb] && !o.isLayerSurfacePoint[nb]) continue;
        const int prev = firstClaim_[nb];
        if (prev >= 0) {
            st(o.normals, p, N3{kGreat, kGreat, kGreat});
            st(o.normals, prev, N3{kGreat, kGreat, kGreat});
            continue;
        }
        o.outerMap[p] = nb;
        st(o.normals, p, ld(o.normals, nb));
        firstClaim_[nb] = p;
    }
}

// OBB.C:370-379, and OBB.C:545-553 which depend on the hop count only (blendWithOrthogonalPoints is called with
// maxLayers + 1, SM.C:2299): tabulated here with the host's pow so that the device needs none
void LayerBuilder::finish() {
    LayerSetup& o = out;
    const int P = t_->nPoints;
    for (int p = 0; p < P; ++p)
        if (isUndef(ld(o.normals, p))) { st(o.normals, p, N3{0.0, 0.0, 0.0}); o.outerMap[p] = -1; }
    const double maxL = double(maxLayers_ + 1), minL = double(minLayers_);
    o.lengthOfHops.assign((size_t)maxIter + 2, 0.0);
    o.blendOfHops.assign((size_t)maxIter + 2, 0.0);
    for (int h = 1; h <= maxIter + 1; ++h) {
        const double capped = (double(h - 1) < maxL) ? double(h - 1) : maxL;   // label maxHops = min(nHops - 1, maxLayers)
        o.lengthOfHops[(size_t)h] = edgeLength_ * std::pow(ratio_, double(int(capped)));
        const double slope = -maxBlend_ / (maxL - minL);
        const double y0 = -slope * maxL;
        const double y = y0 + slope * h;
        const double lo = (y < maxBlend_) ? y : maxBlend_;      // Foam::min
        o.blendOfHops[(size_t)h] = (0.0 > lo) ? 0.0 : lo;       // Foam::max
    }
}

std::string buildLayerSetup(const Topology& t, const uint8_t* internal, const std::vector<LayerPatch>& patches, const double* faceArea,
                            double maxBlend, double edgeLength, double ratio, int minLayers, int maxLayers, LayerSetup& o) {
    LayerBuilder b;
    const std::string err = b.begin(t, internal, patches, faceArea, maxBlend, edgeLength, ratio, minLayers, maxLayers);
    if (!err.empty()) return err;
    for (int iter = 0; iter < b.maxIter; ++iter) b.hopsSweep();
    b.normalsAccumulate();
    b.normalsFinish();
    for (int iter = 1; iter <= b.maxIter; ++iter) b.propagateSweep(iter);
    b.finish();
    o = std::move(b.out);
    return "";
}

}  // namespace smgpu
