// tiles.cpp -- see tiles.hpp.
#include "tiles.hpp"

#include <algorithm>

namespace smgpu {

// ELL row widths: multiples of 4 entries (one ushort4 chunk), at least one chunk so that kernels may read chunk 0 unconditionally
static inline int32_t roundUp4(int32_t v) { return v <= 4 ? 4 : (v + 3) & ~3; }

static inline uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// positions sorted along the Z-curve of the given coordinates (3 per element); ties keep id order
static std::vector<int32_t> mortonOrder(int32_t n, const std::vector<double>& xyz) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], xyz[3 * (size_t)i + a]); hi[a] = std::max(hi[a], xyz[3 * (size_t)i + a]); }
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, hi[a] - lo[a]);
    const double scale = ext > 0.0 ? 2097151.0 / ext : 0.0;   // one isotropic scale: bricks stay cubic in space
    std::vector<std::pair<uint64_t, int32_t>> key((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        uint64_t k = 0;
        for (int a = 0; a < 3; ++a) k |= spread21((uint64_t)((xyz[3 * (size_t)i + a] - lo[a]) * scale)) << a;
        key[(size_t)i] = {k, i};
    }
    std::sort(key.begin(), key.end());
    std::vector<int32_t> order((size_t)n);
    for (int32_t i = 0; i < n; ++i) order[(size_t)i] = key[(size_t)i].second;
    return order;
}

static std::vector<int32_t> naturalOrder(int32_t n) {
    std::vector<int32_t> o((size_t)n);
    for (int32_t i = 0; i < n; ++i) o[(size_t)i] = i;
    return o;
}

std::string GeomTiles::build(const Topology& t, const double* pts, bool morton, int32_t nThreads, int32_t capCells,
                             int32_t capPoints, int32_t capFaces) {
    threads = nThreads;
    if (capCells > threads) capCells = threads;
    const auto& cf = t.cellFacesGeom;
    const auto& fp = t.facePoints;
    if (morton) {
        std::vector<double> cc(3 * (size_t)t.nCells, 0.0);
        for (int32_t c = 0; c < t.nCells; ++c) {
            double s[3] = {0, 0, 0};
            int32_t n = 0;
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t f = cf.val[k] & 0x7fffffff;
                for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j, ++n)
                    for (int a = 0; a < 3; ++a) s[a] += pts[3 * (size_t)fp.val[j] + a];
            }
            for (int a = 0; a < 3; ++a) cc[3 * (size_t)c + a] = n ? s[a] / n : 0.0;
        }
        order = mortonOrder(t.nCells, cc);
    } else order = naturalOrder(t.nCells);
    std::vector<int32_t> stampP((size_t)t.nPoints, -1), stampF((size_t)t.nFaces, -1);
    // pass 1: greedy tile boundaries under the three capacities
    cellBeg.assign(1, 0);
    int32_t tile = 0, nP = 0, nF = 0, nC = 0;
    for (int32_t ci = 0; ci < t.nCells; ++ci) {
        const int32_t c = order[(size_t)ci];
        for (int attempt = 0; attempt < 2; ++attempt) {
            int32_t addF = 0, addP = 0;
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t f = cf.val[k] & 0x7fffffff;
                if (stampF[f] != tile) { stampF[f] = tile; ++addF; }
                for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j)
                    if (stampP[fp.val[j]] != tile) { stampP[fp.val[j]] = tile; ++addP; }
            }
            if (nC > 0 && (nC + 1 > capCells || nP + addP > capPoints || nF + addF > capFaces)) {
                cellBeg.push_back(ci);  // close the tile before this cell and re-add the cell to a fresh one
                ++tile; nP = nF = nC = 0;
                continue;
            }
            if (addP > capPoints || addF > capFaces) return "a single cell exceeds the LDS tile capacity";
            nP += addP; nF += addF; ++nC;
            break;
        }
    }
    cellBeg.push_back(t.nCells);
    nTiles = (int32_t)cellBeg.size() - 1;

    // pass 2: per tile unique lists (ascending), local indices, ELL tables
    std::fill(stampP.begin(), stampP.end(), -1);
    std::fill(stampF.begin(), stampF.end(), -1);
    std::vector<int32_t> locP((size_t)t.nPoints, -1), locF((size_t)t.nFaces, -1);
    tpOff.assign(1, 0); tfOff.assign(1, 0);
    tpIds.clear(); tfIds.clear(); faceVerts.clear(); cellFaces.clear();
    fvBase.clear(); fvWidth.clear(); cfBase.clear(); cfWidth.clear(); tileFlags.clear();
    std::vector<int32_t> faces, points;
    std::vector<int32_t> cellTile((size_t)t.nCells, -1);   // which tile a cell belongs to
    for (int32_t ti = 0; ti < nTiles; ++ti)
        for (int32_t ci = cellBeg[ti]; ci < cellBeg[ti + 1]; ++ci) cellTile[(size_t)order[(size_t)ci]] = ti;
    for (int32_t ti = 0; ti < nTiles; ++ti) {
        faces.clear(); points.clear();
        const int32_t cb = cellBeg[ti], ce = cellBeg[ti + 1];
        int32_t cw = 0, fw = 0;
        for (int32_t ci = cb; ci < ce; ++ci) {
            const int32_t c = order[(size_t)ci];
            cw = std::max(cw, cf.off[c + 1] - cf.off[c]);
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t f = cf.val[k] & 0x7fffffff;
                if (stampF[f] != ti) { stampF[f] = ti; faces.push_back(f); }
            }
        }
        std::sort(faces.begin(), faces.end());
        bool allQuads = true, allHex = true;
        for (int32_t ci = cb; ci < ce; ++ci) { const int32_t c = order[(size_t)ci]; allHex = allHex && (cf.off[c + 1] - cf.off[c] == 6); }
        for (int32_t f : faces) {
            allQuads = allQuads && (fp.off[f + 1] - fp.off[f] == 4);
            fw = std::max(fw, fp.off[f + 1] - fp.off[f]);
            for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j)
                if (stampP[fp.val[j]] != ti) { stampP[fp.val[j]] = ti; points.push_back(fp.val[j]); }
        }
        std::sort(points.begin(), points.end());
        cw = roundUp4(cw); fw = roundUp4(fw);
        if ((int32_t)points.size() > 32767 || (int32_t)faces.size() > 32767 || cw > 252 || fw > 252)
            return "tile too large for the 15-bit local index tables";
        for (size_t i = 0; i < points.size(); ++i) locP[points[i]] = (int32_t)i;
        for (size_t i = 0; i < faces.size(); ++i) locF[faces[i]] = (int32_t)i;
        tpIds.insert(tpIds.end(), points.begin(), points.end());
        tpOff.push_back((int32_t)tpIds.size());
        fvBase.push_back((int32_t)faceVerts.size());
        fvWidth.push_back((uint8_t)fw);
        for (int32_t f : faces) {
            const bool ownerHere = cellTile[(size_t)t.owner[f]] == ti;
            tfIds.push_back(ownerHere ? (int32_t)(0x80000000u | (uint32_t)f) : f);
            const int32_t n = fp.off[f + 1] - fp.off[f];
            for (int32_t j = 0; j < fw; ++j) faceVerts.push_back(j < n ? (uint16_t)locP[fp.val[fp.off[f] + j]] : kEllPad);
        }
        tfOff.push_back((int32_t)tfIds.size());
        cfBase.push_back((int32_t)cellFaces.size());
        cfWidth.push_back((uint8_t)cw);
        tileFlags.push_back((uint8_t)((allQuads ? 1 : 0) | (allHex ? 2 : 0)));
        const size_t base = cellFaces.size();
        cellFaces.resize(base + (size_t)cw * threads, kEllPad);
        for (int32_t ci = cb; ci < ce; ++ci) {
            const int32_t c = order[(size_t)ci];
            const int32_t tl = ci - cb;
            for (int32_t k = cf.off[c], j = 0; k < cf.off[c + 1]; ++k, ++j) {
                const int32_t v = cf.val[k];
                cellFaces[base + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)(locF[v & 0x7fffffff] | (v < 0 ? 0x8000 : 0));
            }
        }
        maxPoints = std::max(maxPoints, (int32_t)points.size());
        maxFaces = std::max(maxFaces, (int32_t)faces.size());
        if (faceVerts.size() > 0x7fffffffu || cellFaces.size() > 0x7fffffffu) return "tile tables exceed int32 addressing";
    }
    return "";
}

// The face corners of a point, (previous vertex, next vertex) per incident face, are the edges of a small graph on the
// point's neighbours (an octahedron for an interior hex point).  The edge-angle filter evaluates a pair of unit vectors
// per neighbour and corner; ordered as Euler trails (each corner starts where the previous one ended, corners flipped as
// needed -- the filter's four cosines are symmetric in the pair) it can carry one neighbour's vectors over from corner to
// corner and evaluates about half of them.  Hierholzer on the graph with the odd-degree vertices paired by virtual edges;
// the virtual edges are dropped from the output, where a new trail starts.  The set of corners is unchanged.
namespace {
struct ChainScratch {
    std::vector<int32_t> verts, deg, adjOff, adjEdge, ea, eb, stackV, stackE, next;
    std::vector<uint8_t> used;
    std::vector<std::pair<int32_t, int32_t>> comp, out;
};
void chainCorners(std::vector<std::pair<int32_t, int32_t>>& corners, ChainScratch& w) {
    const int n = (int)corners.size();
    if (n < 2) return;
    w.verts.clear();
    for (const auto& c : corners) { w.verts.push_back(c.first); w.verts.push_back(c.second); }
    std::sort(w.verts.begin(), w.verts.end());
    w.verts.erase(std::unique(w.verts.begin(), w.verts.end()), w.verts.end());
    const int nv = (int)w.verts.size();
    auto vid = [&](int32_t v) { return (int)(std::lower_bound(w.verts.begin(), w.verts.end(), v) - w.verts.begin()); };
    w.ea.clear(); w.eb.clear();
    for (const auto& c : corners) { w.ea.push_back(vid(c.first)); w.eb.push_back(vid(c.second)); }
    w.deg.assign((size_t)nv, 0);
    for (int e = 0; e < n; ++e) { ++w.deg[(size_t)w.ea[(size_t)e]]; ++w.deg[(size_t)w.eb[(size_t)e]]; }
    int pending = -1;
    for (int v = 0; v < nv; ++v)
        if (w.deg[(size_t)v] & 1) {
            if (pending < 0) pending = v;
            else { w.ea.push_back(pending); w.eb.push_back(v); ++w.deg[(size_t)pending]; ++w.deg[(size_t)v]; pending = -1; }
        }
    const int ne = (int)w.ea.size();
    w.adjOff.assign((size_t)nv + 1, 0);
    for (int e = 0; e < ne; ++e) { ++w.adjOff[(size_t)w.ea[(size_t)e] + 1]; ++w.adjOff[(size_t)w.eb[(size_t)e] + 1]; }
    for (int v = 0; v < nv; ++v) w.adjOff[(size_t)v + 1] += w.adjOff[(size_t)v];
    w.adjEdge.assign((size_t)w.adjOff[(size_t)nv], 0);
    w.next.assign(w.adjOff.begin(), w.adjOff.end() - 1);
    for (int e = 0; e < ne; ++e) { w.adjEdge[(size_t)w.next[(size_t)w.ea[(size_t)e]]++] = e; w.adjEdge[(size_t)w.next[(size_t)w.eb[(size_t)e]]++] = e; }
    w.used.assign((size_t)ne, 0);
    std::vector<int32_t>& next = w.next;   // per vertex: first adjacency entry not yet looked at
    next.assign(w.adjOff.begin(), w.adjOff.end() - 1);
    std::vector<std::pair<int32_t, int32_t>>& out = w.out;
    out.clear();
    for (int start = 0; start < nv; ++start) {
        if (next[(size_t)start] >= w.adjOff[(size_t)start + 1]) continue;
        // Hierholzer: walk until stuck, back up emitting edges; the emitted sequence reversed is a circuit of the component
        w.stackV.assign(1, start); w.stackE.assign(1, -1);
        std::vector<std::pair<int32_t, int32_t>>& comp = w.comp;   // (edge, vertex the edge is entered from) in emission order
        comp.clear();
        while (!w.stackV.empty()) {
            const int v = w.stackV.back();
            int e = -1;
            while (next[(size_t)v] < w.adjOff[(size_t)v + 1]) {
                const int cand = w.adjEdge[(size_t)next[(size_t)v]++];
                if (!w.used[(size_t)cand]) { e = cand; break; }
            }
            if (e >= 0) {
                w.used[(size_t)e] = 1;
                const int to = (w.ea[(size_t)e] == v) ? w.eb[(size_t)e] : w.ea[(size_t)e];
                w.stackV.push_back(to); w.stackE.push_back(e);
            } else {
                const int eIn = w.stackE.back();
                w.stackV.pop_back(); w.stackE.pop_back();
                if (eIn >= 0) comp.push_back({eIn, w.stackV.back()});   // traversed from stackV.back() to v
            }
        }
        // reversed emission order: consecutive edges share a vertex; edge (e, from) runs from `from` to its other end
        for (size_t i = comp.size(); i-- > 0;) {
            const int e = comp[i].first, from = comp[i].second;
            if (e >= n) continue;   // virtual edge: a new trail starts after it
            const int to = (w.ea[(size_t)e] == from) ? w.eb[(size_t)e] : w.ea[(size_t)e];
            out.push_back({w.verts[(size_t)from], w.verts[(size_t)to]});
        }
    }
    if ((int)out.size() == n) corners.swap(out);   // (always; kept as a guard)
}
}  // namespace

std::string SmoothTiles::build(const Topology& t, const double* xyz, const uint8_t* isInternal, bool morton, int32_t nThreads,
                               int32_t capCells, int32_t capPoints) {
    threads = nThreads;
    if (morton) order = mortonOrder(t.nPoints, std::vector<double>(xyz, xyz + 3 * (size_t)t.nPoints));
    else order = naturalOrder(t.nPoints);
    const int32_t capTile = threads;
    const auto& pc = t.pointCells;
    const auto& pe = t.pointEdges;   // offsets shared with pointPoints
    std::vector<int32_t> stampC((size_t)t.nCells, -1), stampN((size_t)t.nPoints, -1);
    ptBeg.assign(1, 0);
    int32_t tile = 0, nC = 0, nN = 0, nT = 0;
    for (int32_t pi = 0; pi < t.nPoints; ++pi) {
        const int32_t p = order[(size_t)pi];
        for (int attempt = 0; attempt < 2; ++attempt) {
            int32_t addC = 0, addN = 0;
            for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k)
                if (stampC[pc.val[k]] != tile) { stampC[pc.val[k]] = tile; ++addC; }
            if (stampN[p] != tile) { stampN[p] = tile; ++addN; }
            for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k)
                if (stampN[t.pointPoints[k]] != tile) { stampN[t.pointPoints[k]] = tile; ++addN; }
            if (nT > 0 && (nT + 1 > capTile || nC + addC > capCells || nN + addN > capPoints)) {
                ptBeg.push_back(pi);
                ++tile; nC = nN = nT = 0;
                continue;
            }
            if (addC > capCells || addN > capPoints) return "a single point exceeds the LDS tile capacity";
            nC += addC; nN += addN; ++nT;
            break;
        }
    }
    ptBeg.push_back(t.nPoints);
    nTiles = (int32_t)ptBeg.size() - 1;

    std::fill(stampC.begin(), stampC.end(), -1);
    std::fill(stampN.begin(), stampN.end(), -1);
    std::vector<int32_t> locC((size_t)t.nCells, -1), locN((size_t)t.nPoints, -1);
    tcOff.assign(1, 0); tnOff.assign(1, 0);
    tcIds.clear(); tnIds.clear();
    selfLoc.assign((size_t)t.nPoints, 0);
    pcBase.clear(); pcWidth.clear(); pcEll.clear(); ppBase.clear(); ppWidth.clear(); ppEll.clear(); pairEll.clear();
    pfBase.clear(); pfWidth.clear(); pfEll.clear();
    const bool pairs = t.maxPointPoints <= 16;
    std::vector<int32_t> cells, pts;
    std::vector<std::pair<int32_t, int32_t>> corners;
    ChainScratch chainScratch;
    for (int32_t ti = 0; ti < nTiles; ++ti) {
        cells.clear(); pts.clear();
        const int32_t pb = ptBeg[ti], pend = ptBeg[ti + 1];
        int32_t wc = 0, wn = 0, wf = 0;
        for (int32_t pi = pb; pi < pend; ++pi) {
            const int32_t p = order[(size_t)pi];
            wf = std::max(wf, 2 * (t.pointFaces.off[p + 1] - t.pointFaces.off[p]));
            wc = std::max(wc, pc.off[p + 1] - pc.off[p]);
            wn = std::max(wn, pe.off[p + 1] - pe.off[p]);
            for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k)
                if (stampC[pc.val[k]] != ti) { stampC[pc.val[k]] = ti; cells.push_back(pc.val[k]); }
            if (stampN[p] != ti) { stampN[p] = ti; pts.push_back(p); }
            for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k)
                if (stampN[t.pointPoints[k]] != ti) { stampN[t.pointPoints[k]] = ti; pts.push_back(t.pointPoints[k]); }
        }
        std::sort(cells.begin(), cells.end());
        std::sort(pts.begin(), pts.end());
        wc = roundUp4(wc); wn = roundUp4(wn); wf = roundUp4(wf);
        if ((int32_t)cells.size() > 32766 || (int32_t)pts.size() > 32766 || wc > 252 || wn > 252 || wf > 252)
            return "tile too large for the 15-bit local index tables";
        for (size_t i = 0; i < cells.size(); ++i) locC[cells[i]] = (int32_t)i;
        for (size_t i = 0; i < pts.size(); ++i) locN[pts[i]] = (int32_t)i;
        tcIds.insert(tcIds.end(), cells.begin(), cells.end());
        tcOff.push_back((int32_t)tcIds.size());
        tnIds.insert(tnIds.end(), pts.begin(), pts.end());
        tnOff.push_back((int32_t)tnIds.size());
        pcBase.push_back((int32_t)pcEll.size()); pcWidth.push_back((uint8_t)wc);
        ppBase.push_back((int32_t)ppEll.size()); ppWidth.push_back((uint8_t)wn);
        const size_t cbase = pcEll.size(), nbase = ppEll.size();
        pcEll.resize(cbase + (size_t)wc * threads, kEllPad);
        ppEll.resize(nbase + (size_t)wn * threads, kEllPad);
        pairEll.resize(nbase + (size_t)wn * threads, 0);
        pfBase.push_back((int32_t)pfEll.size()); pfWidth.push_back((uint8_t)wf);
        const size_t fbase = pfEll.size();
        pfEll.resize(fbase + (size_t)wf * threads, kEllPad);
        for (int32_t pi = pb; pi < pend; ++pi) {
            const int32_t p = order[(size_t)pi];
            const int32_t tl = pi - pb;
            selfLoc[(size_t)pi] = (uint16_t)locN[p];
            corners.clear();
            for (int32_t k = t.pointFaces.off[p]; k < t.pointFaces.off[p + 1]; ++k) corners.push_back({t.pfPrev[k], t.pfNext[k]});
            chainCorners(corners, chainScratch);
            for (int32_t j = 0; j < 2 * (int32_t)corners.size(); j += 2) {
                pfEll[fbase + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)locN[corners[(size_t)j / 2].first];
                pfEll[fbase + ((size_t)((j + 1) / 4) * threads + tl) * 4 + ((j + 1) % 4)] = (uint16_t)locN[corners[(size_t)j / 2].second];
            }
            for (int32_t k = pc.off[p], j = 0; k < pc.off[p + 1]; ++k, ++j)
                pcEll[cbase + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)locC[pc.val[k]];
            const int32_t b = pe.off[p], v = pe.off[p + 1] - b;
            for (int32_t j = 0; j < v; ++j) {
                const int32_t q = t.pointPoints[b + j];
                ppEll[nbase + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)(locN[q] | (isInternal[q] ? 0x8000 : 0));
            }
            if (pairs) {
                // neighbours i, j of p share a cell  <=>  pointCells(q_i) and pointCells(q_j) intersect
                for (int32_t i = 0; i < v; ++i) {
                    const int32_t qi = t.pointPoints[b + i];
                    uint16_t mask = 0;
                    for (int32_t j = 0; j < v; ++j) {
                        if (j == i) continue;
                        const int32_t qj = t.pointPoints[b + j];
                        int32_t a = pc.off[qi], ae = pc.off[qi + 1], c = pc.off[qj], ce = pc.off[qj + 1];
                        while (a < ae && c < ce) {
                            if (pc.val[a] == pc.val[c]) { mask |= (uint16_t)(1u << j); break; }
                            if (pc.val[a] < pc.val[c]) ++a; else ++c;
                        }
                    }
                    pairEll[nbase + ((size_t)(i / 4) * threads + tl) * 4 + (i % 4)] = mask;
                }
            }
        }
        maxCells = std::max(maxCells, (int32_t)cells.size());
        maxPoints = std::max(maxPoints, (int32_t)pts.size());
        if (pcEll.size() > 0x7fffffffu || ppEll.size() > 0x7fffffffu || pfEll.size() > 0x7fffffffu) return "tile tables exceed int32 addressing";
    }
    return "";
}

std::string EdgeTiles::build(const Topology& t, const double* xyz, bool morton, int32_t nThreads, int32_t capPoints,
                             int32_t capFaces, int32_t capCells) {
    threads = nThreads;
    const int32_t nE = t.nEdges;
    if (morton) {
        std::vector<double> mid(3 * (size_t)nE);
        for (int32_t e = 0; e < nE; ++e)
            for (int a = 0; a < 3; ++a) mid[3 * (size_t)e + a] = 0.5 * (xyz[3 * (size_t)t.edges[2 * e] + a] + xyz[3 * (size_t)t.edges[2 * e + 1] + a]);
        order = mortonOrder(nE, mid);
    } else order = naturalOrder(nE);
    const auto& ef = t.edgeFaces;
    const auto& ec = t.edgeCells;
    std::vector<int32_t> stP((size_t)t.nPoints, -1), stF((size_t)t.nFaces, -1), stC((size_t)t.nCells, -1);
    edgeBeg.assign(1, 0);
    int32_t tile = 0, nP = 0, nF = 0, nC = 0, nT = 0;
    for (int32_t ei = 0; ei < nE; ++ei) {
        const int32_t e = order[(size_t)ei];
        for (int attempt = 0; attempt < 2; ++attempt) {
            int32_t aP = 0, aF = 0, aC = 0;
            for (int k = 0; k < 2; ++k) { const int32_t p = t.edges[2 * e + k]; if (stP[p] != tile) { stP[p] = tile; ++aP; } }
            for (int32_t k = ef.off[e]; k < ef.off[e + 1]; ++k) { const int32_t f = ef.val[k]; if (stF[f] != tile) { stF[f] = tile; ++aF; } }
            for (int32_t k = ec.off[e]; k < ec.off[e + 1]; ++k) { const int32_t cI = ec.val[k]; if (stC[cI] != tile) { stC[cI] = tile; ++aC; } }
            if (nT > 0 && (nT + 1 > threads || nP + aP > capPoints || nF + aF > capFaces || nC + aC > capCells)) {
                edgeBeg.push_back(ei);
                ++tile; nP = nF = nC = nT = 0;
                continue;
            }
            if (aP > capPoints || aF > capFaces || aC > capCells) return "a single edge exceeds the LDS tile capacity";
            nP += aP; nF += aF; nC += aC; ++nT;
            break;
        }
    }
    edgeBeg.push_back(nE);
    nTiles = (int32_t)edgeBeg.size() - 1;
    std::fill(stP.begin(), stP.end(), -1); std::fill(stF.begin(), stF.end(), -1); std::fill(stC.begin(), stC.end(), -1);
    std::vector<int32_t> locP((size_t)t.nPoints, -1), locF((size_t)t.nFaces, -1), locC((size_t)t.nCells, -1);
    tpOff.assign(1, 0); tfOff.assign(1, 0); tcOff.assign(1, 0);
    tpIds.clear(); tfIds.clear(); tcIds.clear(); efBase.clear(); ecBase.clear(); efWidth.clear(); ecWidth.clear(); efEll.clear(); ecEll.clear();
    epLoc.assign(2 * (size_t)nE, 0);
    std::vector<int32_t> pts, fcs, cls;
    for (int32_t ti = 0; ti < nTiles; ++ti) {
        pts.clear(); fcs.clear(); cls.clear();
        const int32_t eb = edgeBeg[ti], ee = edgeBeg[ti + 1];
        int32_t wf = 0, wc = 0;
        for (int32_t ei = eb; ei < ee; ++ei) {
            const int32_t e = order[(size_t)ei];
            wf = std::max(wf, ef.off[e + 1] - ef.off[e]);
            wc = std::max(wc, ec.off[e + 1] - ec.off[e]);
            for (int k = 0; k < 2; ++k) { const int32_t p = t.edges[2 * e + k]; if (stP[p] != ti) { stP[p] = ti; pts.push_back(p); } }
            for (int32_t k = ef.off[e]; k < ef.off[e + 1]; ++k) { const int32_t f = ef.val[k]; if (stF[f] != ti) { stF[f] = ti; fcs.push_back(f); } }
            for (int32_t k = ec.off[e]; k < ec.off[e + 1]; ++k) { const int32_t cI = ec.val[k]; if (stC[cI] != ti) { stC[cI] = ti; cls.push_back(cI); } }
        }
        std::sort(pts.begin(), pts.end()); std::sort(fcs.begin(), fcs.end()); std::sort(cls.begin(), cls.end());
        wf = roundUp4(wf); wc = roundUp4(wc);
        if (pts.size() > 32766 || fcs.size() > 32766 || cls.size() > 32766 || wf > 252 || wc > 252)
            return "tile too large for the 15-bit local index tables";
        for (size_t i = 0; i < pts.size(); ++i) locP[pts[i]] = (int32_t)i;
        for (size_t i = 0; i < fcs.size(); ++i) locF[fcs[i]] = (int32_t)i;
        for (size_t i = 0; i < cls.size(); ++i) locC[cls[i]] = (int32_t)i;
        tpIds.insert(tpIds.end(), pts.begin(), pts.end()); tpOff.push_back((int32_t)tpIds.size());
        tfIds.insert(tfIds.end(), fcs.begin(), fcs.end()); tfOff.push_back((int32_t)tfIds.size());
        tcIds.insert(tcIds.end(), cls.begin(), cls.end()); tcOff.push_back((int32_t)tcIds.size());
        efBase.push_back((int32_t)efEll.size()); efWidth.push_back((uint8_t)wf);
        ecBase.push_back((int32_t)ecEll.size()); ecWidth.push_back((uint8_t)wc);
        const size_t fb = efEll.size(), cb = ecEll.size();
        efEll.resize(fb + (size_t)wf * threads, kEllPad);
        ecEll.resize(cb + (size_t)wc * threads, kEllPad);
        for (int32_t ei = eb; ei < ee; ++ei) {
            const int32_t e = order[(size_t)ei], tl = ei - eb;
            epLoc[2 * (size_t)ei] = (uint16_t)locP[t.edges[2 * e]];
            epLoc[2 * (size_t)ei + 1] = (uint16_t)locP[t.edges[2 * e + 1]];
            if (!t.edgeRingOk[(size_t)e]) continue;   // all-pad rows: the kernel flags the edge UNSURE
            for (int32_t k = ef.off[e], j = 0; k < ef.off[e + 1]; ++k, ++j)
                efEll[fb + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)locF[t.ringFace[k]];
            for (int32_t k = ec.off[e], j = 0; k < ec.off[e + 1]; ++k, ++j)
                ecEll[cb + ((size_t)(j / 4) * threads + tl) * 4 + (j % 4)] = (uint16_t)locC[t.ringCell[k]];
        }
        maxPoints = std::max(maxPoints, (int32_t)pts.size());
        maxFaces = std::max(maxFaces, (int32_t)fcs.size());
        maxCells = std::max(maxCells, (int32_t)cls.size());
        if (efEll.size() > 0x7fffffffu || ecEll.size() > 0x7fffffffu) return "tile tables exceed int32 addressing";
    }
    return "";
}

}  // namespace smgpu
