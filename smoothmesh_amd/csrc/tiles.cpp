// tiles.cpp -- see tiles.hpp.
#include "tiles.hpp"
#include "parallel.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace smgpu {

namespace {
struct PhaseTimer {
    const char* who;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit PhaseTimer(const char* w) : who(w), on(std::getenv("SMGPU_VERBOSE") && std::atoi(std::getenv("SMGPU_VERBOSE")) >= 2), t(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "[smgpu] %s tiles: %-18s %.2f s   (done at +%.3f s)\n", who, what, std::chrono::duration<double>(now - t).count(), setupClock());
        t = now;
    }
};
}  // namespace


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
static std::vector<int32_t> mortonOrder(int32_t n, const double* xyz) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    {
        const int parts = rangeParts(n);
        std::vector<double> plo(3 * (size_t)parts, 1e300), phi(3 * (size_t)parts, -1e300);
        parallelRanges(n, parts, [&](int part, int64_t b, int64_t e) {
            double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
            for (int64_t i = b; i < e; ++i)
                for (int a = 0; a < 3; ++a) { l[a] = std::min(l[a], xyz[3 * (size_t)i + a]); h[a] = std::max(h[a], xyz[3 * (size_t)i + a]); }
            for (int a = 0; a < 3; ++a) { plo[3 * (size_t)part + a] = l[a]; phi[3 * (size_t)part + a] = h[a]; }
        });
        for (int part = 0; part < parts; ++part)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], plo[3 * (size_t)part + a]); hi[a] = std::max(hi[a], phi[3 * (size_t)part + a]); }
    }
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, hi[a] - lo[a]);
    const double scale = ext > 0.0 ? 2097151.0 / ext : 0.0;   // one isotropic scale: bricks stay cubic in space
    std::vector<std::pair<uint64_t, int32_t>> key((size_t)n);
    parallelRanges(n, rangeParts(n), [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            uint64_t k = 0;
            for (int a = 0; a < 3; ++a) k |= spread21((uint64_t)((xyz[3 * (size_t)i + a] - lo[a]) * scale)) << a;
            key[(size_t)i] = {k, (int32_t)i};
        }
    });
    // sort by (key, id): the keys' top byte cuts the sequence into 256 buckets (an octant subdivision of the curve) that are
    // sorted side by side -- the same order as one std::sort over all pairs
    if (rangeParts(n) > 1) {
        const int shift = 55;   // keys have 63 bits
        std::vector<size_t> cnt(257, 0);
        for (int32_t i = 0; i < n; ++i) ++cnt[(size_t)(key[(size_t)i].first >> shift) + 1];
        for (int bkt = 0; bkt < 256; ++bkt) cnt[(size_t)bkt + 1] += cnt[(size_t)bkt];
        std::vector<std::pair<uint64_t, int32_t>> tmp((size_t)n);
        {
            std::vector<size_t> cur(cnt.begin(), cnt.end() - 1);
            for (int32_t i = 0; i < n; ++i) tmp[cur[(size_t)(key[(size_t)i].first >> shift)]++] = key[(size_t)i];
        }
        key.swap(tmp);
        std::vector<std::pair<uint64_t, int32_t>>().swap(tmp);
        parallelRanges(256, (int)std::min<unsigned>(hostThreads(), 256u), [&](int, int64_t b, int64_t e) {
            for (int64_t bkt = b; bkt < e; ++bkt) std::sort(key.begin() + (ptrdiff_t)cnt[(size_t)bkt], key.begin() + (ptrdiff_t)cnt[(size_t)bkt + 1]);
        });
    } else std::sort(key.begin(), key.end());
    std::vector<int32_t> order((size_t)n);
    for (int32_t i = 0; i < n; ++i) order[(size_t)i] = key[(size_t)i].second;
    return order;
}

std::vector<int32_t> mortonOrderOf(int32_t n, const double* xyz) { return mortonOrder(n, xyz); }

static std::vector<int32_t> naturalOrder(int32_t n) {
    std::vector<int32_t> o((size_t)n);
    for (int32_t i = 0; i < n; ++i) o[(size_t)i] = i;
    return o;
}

std::string GeomTiles::build(const Topology& t, const double* pts, bool morton, int32_t nThreads, int32_t capCells,
                             int32_t capPoints, int32_t capFaces, int32_t capWeighted, int32_t faceWeight) {
    const std::string e = buildBoundaries(t, pts, morton, nThreads, capCells, capPoints, capFaces, capWeighted, faceWeight);
    return e.empty() ? buildTables(t) : e;
}

// segments of the Z-curve the greedy boundary passes tile side by side on large meshes (SMGPU_TILE_SEGMENTS; every cut costs one
// partial tile and a pair of mesh-sized stamp arrays)
static unsigned tileSegments() {
    static const unsigned n = [] { const char* e = std::getenv("SMGPU_TILE_SEGMENTS"); return e ? std::max(1u, std::min((unsigned)std::atoi(e), 64u)) : 16u; }();
    return n;
}
// The mesh-sized "last tile that saw this element" arrays of the greedy passes: zero pages straight from the OS, as transparent
// huge pages (a fault zeroes 2 MB at memory speed; with 4 KB pages the passes' random accesses took a page fault each, 48 threads
// in the kernel's fault path at once: slower than the value-initialised vectors they replaced, which wrote 200 MB per segment
// before the first cell was looked at).  Tiles are numbered from 1 for the stamps.
struct Stamps {
    int32_t* p;
    size_t bytes;
    explicit Stamps(size_t n) : bytes(std::max<size_t>(n, 1) * sizeof(int32_t)) {
        void* m = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        p = (int32_t*)m;
        adviseHuge(m, bytes);
    }
    ~Stamps() { ::munmap(p, bytes); }
    Stamps(const Stamps&) = delete;
    Stamps& operator=(const Stamps&) = delete;
    int32_t& operator[](size_t i) { return p[i]; }
};

std::string GeomTiles::buildBoundaries(const Topology& t, const double* pts, bool morton, int32_t nThreads, int32_t capCells,
                                       int32_t capPoints, int32_t capFaces, int32_t capWeighted, int32_t faceWeight, const std::vector<int32_t>* cellOrder) {
    threads = nThreads;
    PhaseTimer tm("geometry");
    if (capCells > threads) capCells = threads;
    const auto& cf = t.cellFacesGeom;
    const auto& fp = t.facePoints;
    if (morton && cellOrder && (int64_t)cellOrder->size() == t.nCells) order = *cellOrder;
    else if (morton) {
        std::vector<double> cc(3 * (size_t)t.nCells, 0.0);
        parallelRanges(t.nCells, rangeParts(t.nCells), [&](int, int64_t cb0, int64_t ce0) {
            for (int32_t c = (int32_t)cb0; c < (int32_t)ce0; ++c) {
                double s[3] = {0, 0, 0};
                int32_t n = 0;
                for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                    const int32_t f = cf.val[k] & 0x7fffffff;
                    for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j, ++n)
                        for (int a = 0; a < 3; ++a) s[a] += pts[3 * (size_t)fp.val[j] + a];
                }
                for (int a = 0; a < 3; ++a) cc[3 * (size_t)c + a] = n ? s[a] / n : 0.0;
            }
        });
        order = mortonOrder(t.nCells, cc.data());
    } else order = naturalOrder(t.nCells);
    tm.lap("order");
    // pass 1: greedy tile boundaries under the three capacities (mesh-sized stamp arrays: an L1-resident hash set per open tile
    // was tried in their place and is slower on the hosts of the GPU boxes -- 1.36 s against 0.69 s for 10 M cells)
    // On large meshes the cell sequence is cut into a few segments that are tiled side by side, each with stamp arrays of its own
    // (as the edge tiles do): a segment starts a fresh tile, so the tiling differs from the one-segment tiling by at most one
    // partial tile per cut -- any tiling is as good as any other for the results.  (One pass over 10 M cells: 0.95 s.)
    {
        const int segs = (t.nCells >= (2 << 20)) ? (int)std::min<unsigned>(hostThreads(), tileSegments()) : 1;
        std::vector<std::vector<int32_t>> segBeg((size_t)segs);
        std::vector<std::string> segErr((size_t)segs);
        parallelRanges(t.nCells, segs, [&](int sg, int64_t c0, int64_t c1) {
            Stamps stampP((size_t)t.nPoints), stampF((size_t)t.nFaces);
            std::vector<int32_t>& beg = segBeg[(size_t)sg];
            int32_t tile = 1, nP = 0, nF = 0, nC = 0;
            for (int32_t ci = (int32_t)c0; ci < (int32_t)c1; ++ci) {
                const int32_t c = order[(size_t)ci];
                for (int attempt = 0; attempt < 2; ++attempt) {
                    int32_t addF = 0, addP = 0;
                    for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                        const int32_t f = cf.val[k] & 0x7fffffff;
                        if (stampF[f] != tile) { stampF[f] = tile; ++addF; }
                        for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j)
                            if (stampP[fp.val[j]] != tile) { stampP[fp.val[j]] = tile; ++addP; }
                    }
                    if (nC > 0 && (nC + 1 > capCells || nP + addP > capPoints || nF + addF > capFaces ||
                                   3 * (int64_t)(nP + addP) + (int64_t)faceWeight * (nF + addF) > capWeighted)) {
                        beg.push_back(ci);  // close the tile before this cell and re-add the cell to a fresh one
                        ++tile; nP = nF = nC = 0;
                        continue;
                    }
                    if (addP > capPoints || addF > capFaces) { segErr[(size_t)sg] = "a single cell exceeds the LDS tile capacity"; return; }
                    nP += addP; nF += addF; ++nC;
                    break;
                }
            }
        });
        cellBeg.assign(1, 0);
        for (int sg = 0; sg < segs; ++sg) {
            if (!segErr[(size_t)sg].empty()) return segErr[(size_t)sg];
            if (sg > 0) cellBeg.push_back((int32_t)((int64_t)t.nCells * sg / segs));     // the cut itself
            cellBeg.insert(cellBeg.end(), segBeg[(size_t)sg].begin(), segBeg[(size_t)sg].end());
        }
        cellBeg.push_back(t.nCells);
        nTiles = (int32_t)cellBeg.size() - 1;
    }
    tm.lap("boundaries");
    return "";
}

std::string GeomTiles::buildTables(const Topology& t) {
    PhaseTimer tm("geometry");
    const auto& cf = t.cellFacesGeom;
    const auto& fp = t.facePoints;
    // pass 2: per tile unique lists (ascending), local indices, ELL tables -- tile ranges on host threads, every range
    // builds its share of the tables, the shares are concatenated in tile order (local indices by binary search in the
    // tile's sorted lists: no mesh-sized scratch per thread)
    std::vector<int32_t> cellTile((size_t)t.nCells, -1);   // which tile a cell belongs to
    for (int32_t ti = 0; ti < nTiles; ++ti)
        for (int32_t ci = cellBeg[ti]; ci < cellBeg[ti + 1]; ++ci) cellTile[(size_t)order[(size_t)ci]] = ti;
    struct Part {
        std::vector<int32_t> tpIds, tfIds, nPts, nFcs, fvBase, cfBase;
        std::vector<uint8_t> fvWidth, cfWidth, tileFlags;
        std::vector<uint16_t> faceVerts, cellFaces;
        int32_t maxPoints = 0, maxFaces = 0;
        std::string err;
    };
    const int parts = rangeParts(nTiles, 64);
    std::vector<Part> P((size_t)parts);
    const int32_t threadsL = threads;
    parallelRanges(nTiles, parts, [&](int part, int64_t tb, int64_t te) {
        Part& o = P[(size_t)part];
        std::vector<int32_t> faces, points;
        for (int32_t ti = (int32_t)tb; ti < (int32_t)te; ++ti) {
            faces.clear(); points.clear();
            const int32_t cb = cellBeg[ti], ce = cellBeg[ti + 1];
            int32_t cw = 0, fw = 0;
            bool allQuads = true, allHex = true;
            for (int32_t ci = cb; ci < ce; ++ci) {
                const int32_t c = order[(size_t)ci];
                cw = std::max(cw, cf.off[c + 1] - cf.off[c]);
                allHex = allHex && (cf.off[c + 1] - cf.off[c] == 6);
                for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) faces.push_back(cf.val[k] & 0x7fffffff);
            }
            std::sort(faces.begin(), faces.end());
            faces.erase(std::unique(faces.begin(), faces.end()), faces.end());
            for (int32_t f : faces) {
                allQuads = allQuads && (fp.off[f + 1] - fp.off[f] == 4);
                fw = std::max(fw, fp.off[f + 1] - fp.off[f]);
                for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j) points.push_back(fp.val[j]);
            }
            std::sort(points.begin(), points.end());
            points.erase(std::unique(points.begin(), points.end()), points.end());
            cw = roundUp4(cw); fw = roundUp4(fw);
            if ((int32_t)points.size() > 32767 || (int32_t)faces.size() > 32767 || cw > 252 || fw > 252) {
                o.err = "tile too large for the 15-bit local index tables";
                return;
            }
            auto locP = [&](int32_t p) { return (int32_t)(std::lower_bound(points.begin(), points.end(), p) - points.begin()); };
            auto locF = [&](int32_t f) { return (int32_t)(std::lower_bound(faces.begin(), faces.end(), f) - faces.begin()); };
            o.tpIds.insert(o.tpIds.end(), points.begin(), points.end());
            o.nPts.push_back((int32_t)points.size());
            o.fvBase.push_back((int32_t)o.faceVerts.size());
            o.fvWidth.push_back((uint8_t)fw);
            for (int32_t f : faces) {
                const bool ownerHere = cellTile[(size_t)t.owner[f]] == ti;
                o.tfIds.push_back(ownerHere ? (int32_t)(0x80000000u | (uint32_t)f) : f);
                const int32_t n = fp.off[f + 1] - fp.off[f];
                for (int32_t j = 0; j < fw; ++j) o.faceVerts.push_back(j < n ? (uint16_t)locP(fp.val[fp.off[f] + j]) : kEllPad);
            }
            o.nFcs.push_back((int32_t)faces.size());
            o.cfBase.push_back((int32_t)o.cellFaces.size());
            o.cfWidth.push_back((uint8_t)cw);
            o.tileFlags.push_back((uint8_t)((allQuads ? 1 : 0) | (allHex ? 2 : 0)));
            const size_t base = o.cellFaces.size();
            o.cellFaces.resize(base + (size_t)cw * threadsL, kEllPad);
            for (int32_t ci = cb; ci < ce; ++ci) {
                const int32_t c = order[(size_t)ci];
                const int32_t tl = ci - cb;
                for (int32_t k = cf.off[c], j = 0; k < cf.off[c + 1]; ++k, ++j) {
                    const int32_t v = cf.val[k];
                    o.cellFaces[base + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)(locF(v & 0x7fffffff) | (v < 0 ? 0x8000 : 0));
                }
            }
            o.maxPoints = std::max(o.maxPoints, (int32_t)points.size());
            o.maxFaces = std::max(o.maxFaces, (int32_t)faces.size());
        }
    });
    tpOff.assign(1, 0); tfOff.assign(1, 0);
    fvBase.clear(); fvWidth.clear(); cfBase.clear(); cfWidth.clear(); tileFlags.clear();
    size_t fvTotal = 0, cfTotal = 0;
    for (Part& o : P) {
        if (!o.err.empty()) return o.err;
        for (int32_t n : o.nPts) tpOff.push_back(tpOff.back() + n);
        for (int32_t n : o.nFcs) tfOff.push_back(tfOff.back() + n);
        for (int32_t b : o.fvBase) fvBase.push_back((int32_t)(fvTotal + (size_t)b));
        for (int32_t b : o.cfBase) cfBase.push_back((int32_t)(cfTotal + (size_t)b));
        fvWidth.insert(fvWidth.end(), o.fvWidth.begin(), o.fvWidth.end());
        cfWidth.insert(cfWidth.end(), o.cfWidth.begin(), o.cfWidth.end());
        tileFlags.insert(tileFlags.end(), o.tileFlags.begin(), o.tileFlags.end());
        fvTotal += o.faceVerts.size(); cfTotal += o.cellFaces.size();
        maxPoints = std::max(maxPoints, o.maxPoints);
        maxFaces = std::max(maxFaces, o.maxFaces);
    }
    if (fvTotal > 0x7fffffffu || cfTotal > 0x7fffffffu) return "tile tables exceed int32 addressing";
    {
        std::vector<std::vector<int32_t>> a, b;
        std::vector<std::vector<uint16_t>> c, d;
        for (Part& o : P) { a.push_back(std::move(o.tpIds)); b.push_back(std::move(o.tfIds)); c.push_back(std::move(o.faceVerts)); d.push_back(std::move(o.cellFaces)); }
        concatParts(tpIds, a); concatParts(tfIds, b); concatParts(faceVerts, c); concatParts(cellFaces, d);
    }
    tm.lap("tables");
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
                               int32_t capCells, int32_t capPoints, const std::vector<int32_t>* pointOrder, const std::vector<int32_t>* subset, int32_t capTotal) {
    const std::string e = buildBoundaries(t, xyz, morton, nThreads, capCells, capPoints, pointOrder, subset, capTotal);
    return e.empty() ? buildTables(t, isInternal, subset != nullptr) : e;
}

std::string SmoothTiles::buildBoundaries(const Topology& t, const double* xyz, bool morton, int32_t nThreads, int32_t capCells,
                                         int32_t capPoints, const std::vector<int32_t>* pointOrder, const std::vector<int32_t>* subset, int32_t capTotal) {
    threads = nThreads;
    PhaseTimer tm(subset ? "shared-point" : "smoothing");
    if (subset) order = *subset;
    else if (morton) order = pointOrder ? *pointOrder : mortonOrder(t.nPoints, xyz);
    else order = naturalOrder(t.nPoints);
    const int32_t nPos = (int32_t)order.size();      // positions = points to tile (all of them, or the subset)
    maxCells = maxPoints = 0;
    tm.lap("order");
    const int32_t capTile = threads;
    const auto& pc = t.pointCells;
    const auto& pe = t.pointEdges;   // offsets shared with pointPoints
    // (segments tiled side by side on large point sets, see GeomTiles::build)
    {
        const int segs = (nPos >= (2 << 20)) ? (int)std::min<unsigned>(hostThreads(), tileSegments()) : 1;
        std::vector<std::vector<int32_t>> segBeg((size_t)segs);
        std::vector<std::string> segErr((size_t)segs);
        parallelRanges(nPos, segs, [&](int sg, int64_t p0, int64_t p1) {
            Stamps stampC((size_t)t.nCells), stampN((size_t)t.nPoints);
            std::vector<int32_t>& beg = segBeg[(size_t)sg];
            int32_t tile = 1, nC = 0, nN = 0, nT = 0;
            for (int32_t pi = (int32_t)p0; pi < (int32_t)p1; ++pi) {
                const int32_t p = order[(size_t)pi];
                for (int attempt = 0; attempt < 2; ++attempt) {
                    int32_t addC = 0, addN = 0;
                    for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k)
                        if (stampC[pc.val[k]] != tile) { stampC[pc.val[k]] = tile; ++addC; }
                    if (stampN[p] != tile) { stampN[p] = tile; ++addN; }
                    for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k)
                        if (stampN[t.pointPoints[k]] != tile) { stampN[t.pointPoints[k]] = tile; ++addN; }
                    if (nT > 0 && (nT + 1 > capTile || nC + addC > capCells || nN + addN > capPoints || nC + addC + nN + addN > capTotal)) {
                        beg.push_back(pi);
                        ++tile; nC = nN = nT = 0;
                        continue;
                    }
                    if (addC > capCells || addN > capPoints) { segErr[(size_t)sg] = "a single point exceeds the LDS tile capacity"; return; }
                    nC += addC; nN += addN; ++nT;
                    break;
                }
            }
        });
        ptBeg.assign(1, 0);
        for (int sg = 0; sg < segs; ++sg) {
            if (!segErr[(size_t)sg].empty()) return segErr[(size_t)sg];
            if (sg > 0) ptBeg.push_back((int32_t)((int64_t)nPos * sg / segs));     // the cut itself
            ptBeg.insert(ptBeg.end(), segBeg[(size_t)sg].begin(), segBeg[(size_t)sg].end());
        }
        ptBeg.push_back(nPos);
        nTiles = (nPos > 0) ? (int32_t)ptBeg.size() - 1 : 0;
        if (nPos == 0) ptBeg.assign(1, 0);
    }
    tm.lap("boundaries");
    return "";
}

std::string SmoothTiles::buildTables(const Topology& t, const uint8_t* isInternal, bool subset) {
    PhaseTimer tm(subset ? "shared-point" : "smoothing");
    const int32_t nPos = (int32_t)order.size();
    const auto& pc = t.pointCells;
    const auto& pe = t.pointEdges;   // offsets shared with pointPoints
    selfLoc.assign((size_t)nPos, 0);
    const bool pairs = t.maxPointPoints <= 16;
    struct Part {
        std::vector<int32_t> tcIds, tnIds, nCl, nPt, pcBase, ppBase, pfBase;
        std::vector<uint8_t> pcWidth, ppWidth, pfWidth;
        std::vector<uint16_t> pcEll, ppEll, pairEll, pfEll;
        int32_t maxCells = 0, maxPoints = 0;
        std::string err;
    };
    const int parts = rangeParts(nTiles, 64);
    std::vector<Part> P((size_t)parts);
    const int32_t threadsL = threads;
    parallelRanges(nTiles, parts, [&](int part, int64_t tb, int64_t te) {
        Part& o = P[(size_t)part];
        std::vector<int32_t> cells, pts;
        std::vector<std::pair<int32_t, int32_t>> corners;
        ChainScratch chainScratch;
        for (int32_t ti = (int32_t)tb; ti < (int32_t)te; ++ti) {
            cells.clear(); pts.clear();
            const int32_t pb = ptBeg[ti], pend = ptBeg[ti + 1];
            int32_t wc = 0, wn = 0, wf = 0;
            for (int32_t pi = pb; pi < pend; ++pi) {
                const int32_t p = order[(size_t)pi];
                wf = std::max(wf, 2 * (t.pointFaces.off[p + 1] - t.pointFaces.off[p]));
                wc = std::max(wc, pc.off[p + 1] - pc.off[p]);
                wn = std::max(wn, pe.off[p + 1] - pe.off[p]);
                for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k) cells.push_back(pc.val[k]);
                pts.push_back(p);
                for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k) pts.push_back(t.pointPoints[k]);
            }
            std::sort(cells.begin(), cells.end());
            cells.erase(std::unique(cells.begin(), cells.end()), cells.end());
            std::sort(pts.begin(), pts.end());
            pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
            wc = roundUp4(wc); wn = roundUp4(wn); wf = roundUp4(wf);
            if ((int32_t)cells.size() > 32766 || (int32_t)pts.size() > 32766 || wc > 252 || wn > 252 || wf > 252) {
                o.err = "tile too large for the 15-bit local index tables";
                return;
            }
            auto locC = [&](int32_t c) { return (int32_t)(std::lower_bound(cells.begin(), cells.end(), c) - cells.begin()); };
            auto locN = [&](int32_t q) { return (int32_t)(std::lower_bound(pts.begin(), pts.end(), q) - pts.begin()); };
            o.tcIds.insert(o.tcIds.end(), cells.begin(), cells.end());
            o.nCl.push_back((int32_t)cells.size());
            o.tnIds.insert(o.tnIds.end(), pts.begin(), pts.end());
            o.nPt.push_back((int32_t)pts.size());
            o.pcBase.push_back((int32_t)o.pcEll.size()); o.pcWidth.push_back((uint8_t)wc);
            o.ppBase.push_back((int32_t)o.ppEll.size()); o.ppWidth.push_back((uint8_t)wn);
            const size_t cbase = o.pcEll.size(), nbase = o.ppEll.size();
            o.pcEll.resize(cbase + (size_t)wc * threadsL, kEllPad);
            o.ppEll.resize(nbase + (size_t)wn * threadsL, kEllPad);
            o.pairEll.resize(nbase + (size_t)wn * threadsL, 0);
            o.pfBase.push_back((int32_t)o.pfEll.size()); o.pfWidth.push_back((uint8_t)wf);
            const size_t fbase = o.pfEll.size();
            o.pfEll.resize(fbase + (size_t)wf * threadsL, kEllPad);
            for (int32_t pi = pb; pi < pend; ++pi) {
                const int32_t p = order[(size_t)pi];
                const int32_t tl = pi - pb;
                selfLoc[(size_t)pi] = (uint16_t)locN(p);
                corners.clear();
                for (int32_t k = t.pointFaces.off[p]; k < t.pointFaces.off[p + 1]; ++k) corners.push_back({t.pfPrev[k], t.pfNext[k]});
                chainCorners(corners, chainScratch);
                for (int32_t j = 0; j < 2 * (int32_t)corners.size(); j += 2) {
                    o.pfEll[fbase + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)locN(corners[(size_t)j / 2].first);
                    o.pfEll[fbase + ((size_t)((j + 1) / 4) * threadsL + tl) * 4 + ((j + 1) % 4)] = (uint16_t)locN(corners[(size_t)j / 2].second);
                }
                for (int32_t k = pc.off[p], j = 0; k < pc.off[p + 1]; ++k, ++j)
                    o.pcEll[cbase + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)locC(pc.val[k]);
                const int32_t b = pe.off[p], v = pe.off[p + 1] - b;
                for (int32_t j = 0; j < v; ++j) {
                    const int32_t q = t.pointPoints[b + j];
                    o.ppEll[nbase + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)(locN(q) | (isInternal[q] ? 0x8000 : 0));
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
                        o.pairEll[nbase + ((size_t)(i / 4) * threadsL + tl) * 4 + (i % 4)] = mask;
                    }
                }
            }
            o.maxCells = std::max(o.maxCells, (int32_t)cells.size());
            o.maxPoints = std::max(o.maxPoints, (int32_t)pts.size());
        }
    });
    tcOff.assign(1, 0); tnOff.assign(1, 0);
    pcBase.clear(); pcWidth.clear(); ppBase.clear(); ppWidth.clear(); pfBase.clear(); pfWidth.clear();
    size_t pcTotal = 0, ppTotal = 0, pfTotal = 0;
    for (Part& o : P) {
        if (!o.err.empty()) return o.err;
        for (int32_t n : o.nCl) tcOff.push_back(tcOff.back() + n);
        for (int32_t n : o.nPt) tnOff.push_back(tnOff.back() + n);
        for (int32_t b : o.pcBase) pcBase.push_back((int32_t)(pcTotal + (size_t)b));
        for (int32_t b : o.ppBase) ppBase.push_back((int32_t)(ppTotal + (size_t)b));
        for (int32_t b : o.pfBase) pfBase.push_back((int32_t)(pfTotal + (size_t)b));
        pcWidth.insert(pcWidth.end(), o.pcWidth.begin(), o.pcWidth.end());
        ppWidth.insert(ppWidth.end(), o.ppWidth.begin(), o.ppWidth.end());
        pfWidth.insert(pfWidth.end(), o.pfWidth.begin(), o.pfWidth.end());
        pcTotal += o.pcEll.size(); ppTotal += o.ppEll.size(); pfTotal += o.pfEll.size();
        maxCells = std::max(maxCells, o.maxCells);
        maxPoints = std::max(maxPoints, o.maxPoints);
    }
    if (pcTotal > 0x7fffffffu || ppTotal > 0x7fffffffu || pfTotal > 0x7fffffffu) return "tile tables exceed int32 addressing";
    {
        std::vector<std::vector<int32_t>> a, b;
        std::vector<std::vector<uint16_t>> c, d, e, f;
        for (Part& o : P) {
            a.push_back(std::move(o.tcIds)); b.push_back(std::move(o.tnIds)); c.push_back(std::move(o.pcEll)); d.push_back(std::move(o.ppEll));
            e.push_back(std::move(o.pairEll)); f.push_back(std::move(o.pfEll));
        }
        concatParts(tcIds, a); concatParts(tnIds, b); concatParts(pcEll, c); concatParts(ppEll, d); concatParts(pairEll, e); concatParts(pfEll, f);
    }
    tm.lap("tables");
    return "";
}

std::string EdgeTiles::build(const Topology& t, const double* xyz, bool morton, int32_t nThreads, int32_t capPoints,
                             int32_t capFaces, int32_t capCells, const std::vector<int32_t>* pointOrder, int32_t capTotal) {
    const std::string e = buildBoundaries(t, xyz, morton, nThreads, capPoints, capFaces, capCells, pointOrder, capTotal);
    return e.empty() ? buildTables(t) : e;
}

std::string EdgeTiles::buildBoundaries(const Topology& t, const double* xyz, bool morton, int32_t nThreads, int32_t capPoints,
                                       int32_t capFaces, int32_t capCells, const std::vector<int32_t>* pointOrder, int32_t capTotal) {
    threads = nThreads;
    PhaseTimer tm("edge");
    const int32_t nE = t.nEdges;
    if (morton && pointOrder) {
        // edges are stored in upper-triangular order, i.e. grouped by their start point: walk the points along their Z-curve
        // (startOff[p] = the first edge whose start point is >= p: the edges are sorted by start point, so every stretch of them fills
        // the entries of the points it passes)
        std::vector<int32_t> startOff((size_t)t.nPoints + 1);
        parallelRanges(nE, rangeParts(nE), [&](int, int64_t b, int64_t e1) {
            for (int64_t e = b; e < e1; ++e) {
                const int32_t p = t.edges[2 * e], prev = e > 0 ? t.edges[2 * (e - 1)] : -1;
                for (int32_t q = prev + 1; q <= p; ++q) startOff[(size_t)q] = (int32_t)e;
            }
        });
        for (int32_t q = (nE > 0 ? t.edges[2 * ((size_t)nE - 1)] : -1) + 1; q <= t.nPoints; ++q) startOff[(size_t)q] = nE;
        // (where every point's block of edges starts in the order: a prefix sum along the Z-curve; the blocks are then written side by side)
        const std::vector<int32_t>& po = *pointOrder;
        std::vector<int32_t> outOff(po.size() + 1, 0);
        for (size_t i = 0; i < po.size(); ++i) outOff[i + 1] = outOff[i] + (startOff[(size_t)po[i] + 1] - startOff[(size_t)po[i]]);
        resizeHuge(order, (size_t)nE);
        parallelRanges((int64_t)po.size(), rangeParts((int64_t)po.size()), [&](int, int64_t b, int64_t e1) {
            for (int64_t i = b; i < e1; ++i) {
                int32_t o = outOff[(size_t)i];
                for (int32_t e = startOff[(size_t)po[(size_t)i]]; e < startOff[(size_t)po[(size_t)i] + 1]; ++e) order[(size_t)o++] = e;
            }
        });
    } else if (morton) {
        std::vector<double> mid(3 * (size_t)nE);
        parallelRanges(nE, rangeParts(nE), [&](int, int64_t b, int64_t e1) {
            for (int64_t e = b; e < e1; ++e)
                for (int a = 0; a < 3; ++a) mid[3 * (size_t)e + a] = 0.5 * (xyz[3 * (size_t)t.edges[2 * e] + a] + xyz[3 * (size_t)t.edges[2 * e + 1] + a]);
        });
        order = mortonOrder(nE, mid.data());
    } else order = naturalOrder(nE);
    tm.lap("order");
    const auto& ef = t.edgeFaces;
    const auto& ec = t.edgeCells;
    // pass 1: greedy tile boundaries.  On large meshes the edge sequence is cut into a few segments that are tiled side by side
    // (each with stamp arrays of its own); a segment starts a fresh tile, so the tiling differs from the one-segment tiling by
    // at most one partial tile per cut -- any tiling is as good as any other for the results.
    const int segs = (nE >= (4 << 20)) ? (int)std::min<unsigned>(hostThreads(), tileSegments()) : 1;
    std::vector<std::vector<int32_t>> segBeg((size_t)segs);
    std::vector<std::string> segErr((size_t)segs);
    parallelRanges(nE, segs, [&](int sg, int64_t e0, int64_t e1) {
        Stamps stP((size_t)t.nPoints), stF((size_t)t.nFaces), stC((size_t)t.nCells);
        std::vector<int32_t>& beg = segBeg[(size_t)sg];
        int32_t tile = 1, nP = 0, nF = 0, nC = 0, nT = 0;
        for (int32_t ei = (int32_t)e0; ei < (int32_t)e1; ++ei) {
            const int32_t e = order[(size_t)ei];
            for (int attempt = 0; attempt < 2; ++attempt) {
                int32_t aP = 0, aF = 0, aC = 0;
                for (int k = 0; k < 2; ++k) { const int32_t p = t.edges[2 * e + k]; if (stP[p] != tile) { stP[p] = tile; ++aP; } }
                for (int32_t k = ef.off[e]; k < ef.off[e + 1]; ++k) { const int32_t f = ef.val[k]; if (stF[f] != tile) { stF[f] = tile; ++aF; } }
                for (int32_t k = ec.off[e]; k < ec.off[e + 1]; ++k) { const int32_t cI = ec.val[k]; if (stC[cI] != tile) { stC[cI] = tile; ++aC; } }
                if (nT > 0 && (nT + 1 > threads || nP + aP > capPoints || nF + aF > capFaces || nC + aC > capCells || nP + aP + nF + aF + nC + aC > capTotal)) {
                    beg.push_back(ei);
                    ++tile; nP = nF = nC = nT = 0;
                    continue;
                }
                if (aP > capPoints || aF > capFaces || aC > capCells) { segErr[(size_t)sg] = "a single edge exceeds the LDS tile capacity"; return; }
                nP += aP; nF += aF; nC += aC; ++nT;
                break;
            }
        }
    });
    edgeBeg.assign(1, 0);
    for (int sg = 0; sg < segs; ++sg) {
        if (!segErr[(size_t)sg].empty()) return segErr[(size_t)sg];
        if (sg > 0) edgeBeg.push_back((int32_t)((int64_t)nE * sg / segs));     // the cut itself
        edgeBeg.insert(edgeBeg.end(), segBeg[(size_t)sg].begin(), segBeg[(size_t)sg].end());
    }
    edgeBeg.push_back(nE);
    nTiles = (int32_t)edgeBeg.size() - 1;
    tm.lap("boundaries");
    return "";
}

std::string EdgeTiles::buildTables(const Topology& t) {
    PhaseTimer tm("edge");
    const int32_t nE = t.nEdges;
    const auto& ef = t.edgeFaces;
    const auto& ec = t.edgeCells;
    epLoc.assign(2 * (size_t)nE, 0);
    struct Part {
        std::vector<int32_t> tpIds, tfIds, tcIds, nP, nF, nC, efBase, ecBase;
        std::vector<uint8_t> efWidth, ecWidth;
        std::vector<uint16_t> efEll, ecEll;
        int32_t maxPoints = 0, maxFaces = 0, maxCells = 0;
        std::string err;
    };
    const int parts = rangeParts(nTiles, 64);
    std::vector<Part> P((size_t)parts);
    const int32_t threadsL = threads;
    parallelRanges(nTiles, parts, [&](int part, int64_t tb, int64_t te) {
        Part& o = P[(size_t)part];
        std::vector<int32_t> pts, fcs, cls;
        for (int32_t ti = (int32_t)tb; ti < (int32_t)te; ++ti) {
            pts.clear(); fcs.clear(); cls.clear();
            const int32_t eb = edgeBeg[ti], ee = edgeBeg[ti + 1];
            int32_t wf = 0, wc = 0;
            for (int32_t ei = eb; ei < ee; ++ei) {
                const int32_t e = order[(size_t)ei];
                wf = std::max(wf, ef.off[e + 1] - ef.off[e]);
                wc = std::max(wc, ec.off[e + 1] - ec.off[e]);
                pts.push_back(t.edges[2 * e]); pts.push_back(t.edges[2 * e + 1]);
                for (int32_t k = ef.off[e]; k < ef.off[e + 1]; ++k) fcs.push_back(ef.val[k]);
                for (int32_t k = ec.off[e]; k < ec.off[e + 1]; ++k) cls.push_back(ec.val[k]);
            }
            std::sort(pts.begin(), pts.end()); pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
            std::sort(fcs.begin(), fcs.end()); fcs.erase(std::unique(fcs.begin(), fcs.end()), fcs.end());
            std::sort(cls.begin(), cls.end()); cls.erase(std::unique(cls.begin(), cls.end()), cls.end());
            wf = roundUp4(wf); wc = roundUp4(wc);
            if (pts.size() > 32766 || fcs.size() > 32766 || cls.size() > 32766 || wf > 252 || wc > 252) {
                o.err = "tile too large for the 15-bit local index tables";
                return;
            }
            auto loc = [](const std::vector<int32_t>& v, int32_t x) { return (int32_t)(std::lower_bound(v.begin(), v.end(), x) - v.begin()); };
            o.tpIds.insert(o.tpIds.end(), pts.begin(), pts.end()); o.nP.push_back((int32_t)pts.size());
            o.tfIds.insert(o.tfIds.end(), fcs.begin(), fcs.end()); o.nF.push_back((int32_t)fcs.size());
            o.tcIds.insert(o.tcIds.end(), cls.begin(), cls.end()); o.nC.push_back((int32_t)cls.size());
            o.efBase.push_back((int32_t)o.efEll.size()); o.efWidth.push_back((uint8_t)wf);
            o.ecBase.push_back((int32_t)o.ecEll.size()); o.ecWidth.push_back((uint8_t)wc);
            const size_t fb = o.efEll.size(), cb = o.ecEll.size();
            o.efEll.resize(fb + (size_t)wf * threadsL, kEllPad);
            o.ecEll.resize(cb + (size_t)wc * threadsL, kEllPad);
            for (int32_t ei = eb; ei < ee; ++ei) {
                const int32_t e = order[(size_t)ei], tl = ei - eb;
                epLoc[2 * (size_t)ei] = (uint16_t)loc(pts, t.edges[2 * e]);
                epLoc[2 * (size_t)ei + 1] = (uint16_t)loc(pts, t.edges[2 * e + 1]);
                if (!t.edgeRingOk[(size_t)e]) continue;   // all-pad rows: the kernel flags the edge UNSURE
                for (int32_t k = ef.off[e], j = 0; k < ef.off[e + 1]; ++k, ++j)
                    o.efEll[fb + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)loc(fcs, t.ringFace[k]);
                for (int32_t k = ec.off[e], j = 0; k < ec.off[e + 1]; ++k, ++j)
                    o.ecEll[cb + ((size_t)(j / 4) * threadsL + tl) * 4 + (j % 4)] = (uint16_t)loc(cls, t.ringCell[k]);
            }
            o.maxPoints = std::max(o.maxPoints, (int32_t)pts.size());
            o.maxFaces = std::max(o.maxFaces, (int32_t)fcs.size());
            o.maxCells = std::max(o.maxCells, (int32_t)cls.size());
        }
    });
    tpOff.assign(1, 0); tfOff.assign(1, 0); tcOff.assign(1, 0);
    efBase.clear(); ecBase.clear(); efWidth.clear(); ecWidth.clear();
    size_t efTotal = 0, ecTotal = 0;
    for (Part& o : P) {
        if (!o.err.empty()) return o.err;
        for (int32_t n : o.nP) tpOff.push_back(tpOff.back() + n);
        for (int32_t n : o.nF) tfOff.push_back(tfOff.back() + n);
        for (int32_t n : o.nC) tcOff.push_back(tcOff.back() + n);
        for (int32_t b : o.efBase) efBase.push_back((int32_t)(efTotal + (size_t)b));
        for (int32_t b : o.ecBase) ecBase.push_back((int32_t)(ecTotal + (size_t)b));
        efWidth.insert(efWidth.end(), o.efWidth.begin(), o.efWidth.end());
        ecWidth.insert(ecWidth.end(), o.ecWidth.begin(), o.ecWidth.end());
        efTotal += o.efEll.size(); ecTotal += o.ecEll.size();
        maxPoints = std::max(maxPoints, o.maxPoints);
        maxFaces = std::max(maxFaces, o.maxFaces);
        maxCells = std::max(maxCells, o.maxCells);
    }
    if (efTotal > 0x7fffffffu || ecTotal > 0x7fffffffu) return "tile tables exceed int32 addressing";
    {
        std::vector<std::vector<int32_t>> a, b, c;
        std::vector<std::vector<uint16_t>> d, e;
        for (Part& o : P) {
            a.push_back(std::move(o.tpIds)); b.push_back(std::move(o.tfIds)); c.push_back(std::move(o.tcIds));
            d.push_back(std::move(o.efEll)); e.push_back(std::move(o.ecEll));
        }
        concatParts(tpIds, a); concatParts(tfIds, b); concatParts(tcIds, c); concatParts(efEll, d); concatParts(ecEll, e);
    }
    tm.lap("tables");
    return "";
}

}  // namespace smgpu
