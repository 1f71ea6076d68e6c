// topology.cpp -- see topology.hpp.  Count / scan / fill CSR construction, O(nnz).
#include "topology.hpp"
#include "parallel.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace smgpu {

static void scanCounts(std::vector<int32_t>& off) {
    // off[i+1] holds the count of row i on entry; exclusive scan in place
    int64_t run = 0;
    for (size_t i = 0; i + 1 < off.size(); ++i) {
        const int64_t c = off[i + 1];
        off[i + 1] = (int32_t)(run + c);
        run += c;
    }
}

std::string Topology::build(int32_t nP, int32_t nC, int32_t nF, int32_t nIF, const int32_t* faceOffsets,
                            const int32_t* facePts, const int32_t* own, const int32_t* nei, const std::function<void()>& afterCells,
                            const std::function<void()>& afterPoints) {
    nPoints = nP; nCells = nC; nFaces = nF; nInternalFaces = nIF;
    const bool verbose = std::getenv("SMGPU_VERBOSE") && std::atoi(std::getenv("SMGPU_VERBOSE")) >= 2;
    auto tLap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "[smgpu] addressing: %-24s %.2f s\n", what, std::chrono::duration<double>(now - tLap).count());
        tLap = now;
    };
    if (nP <= 0 || nC <= 0 || nF <= 0 || nIF < 0 || nIF > nF) return "invalid mesh sizes";
    const int64_t nnzFP = faceOffsets[nF];
    if (nnzFP >= (int64_t)1 << 31) return "face-point list exceeds int32 addressing";
    facePoints.off.assign(faceOffsets, faceOffsets + nF + 1);
    facePoints.val.assign(facePts, facePts + nnzFP);
    owner.assign(own, own + nF);
    neighbour.assign(nei, nei + nIF);
    {
        const int parts = rangeParts(nF);
        std::vector<std::string> perr((size_t)parts);
        std::vector<int32_t> pmax((size_t)parts, 0);
        parallelRanges(nF, parts, [&](int part, int64_t fb0, int64_t fe0) {
            int32_t localMax = 0;
            for (int32_t f = (int32_t)fb0; f < (int32_t)fe0; ++f) {
                const int32_t n = faceOffsets[f + 1] - faceOffsets[f];
                std::string e;
                if (n < 3) e = "face " + std::to_string(f) + " has fewer than 3 points";
                else if (own[f] < 0 || own[f] >= nC) e = "owner out of range at face " + std::to_string(f);
                else if (f < nIF && (nei[f] < 0 || nei[f] >= nC)) e = "neighbour out of range at face " + std::to_string(f);
                else
                    for (int32_t k = faceOffsets[f]; k < faceOffsets[f + 1]; ++k)
                        if (facePts[k] < 0 || facePts[k] >= nP) { e = "face point label out of range"; break; }
                if (!e.empty()) { perr[(size_t)part] = e; return; }
                localMax = std::max(localMax, n);
            }
            pmax[(size_t)part] = localMax;
        });
        for (const std::string& e : perr) if (!e.empty()) return e;     // (the first failing face in face order)
        for (int32_t v : pmax) maxFaceSize = std::max(maxFaceSize, v);
    }

    lap("copy + checks");
    // ---- cell -> faces (geometry accumulation order) -------------------------------------
    cellFacesGeom.off.assign(nC + 1, 0);
    for (int32_t f = 0; f < nF; ++f) cellFacesGeom.off[own[f] + 1]++;
    for (int32_t f = 0; f < nIF; ++f) cellFacesGeom.off[nei[f] + 1]++;
    scanCounts(cellFacesGeom.off);
    cellFacesGeom.val.resize(cellFacesGeom.off[nC]);
    {
        std::vector<int32_t> cur(cellFacesGeom.off.begin(), cellFacesGeom.off.end() - 1);
        for (int32_t f = 0; f < nF; ++f) cellFacesGeom.val[cur[own[f]]++] = f;
        for (int32_t f = 0; f < nIF; ++f) cellFacesGeom.val[cur[nei[f]]++] = (int32_t)(0x80000000u | (uint32_t)f);
    }

    lap("cellFaces");
    if (afterCells) afterCells();
    // ---- pointFaces with prev/next vertex -------------------------------------------------
    pointFaces.off.assign(nP + 1, 0);
    for (int64_t i = 0; i < nnzFP; ++i) pointFaces.off[facePts[i] + 1]++;
    scanCounts(pointFaces.off);
    pointFaces.val.resize(nnzFP);
    pfPrev.resize(nnzFP);
    pfNext.resize(nnzFP);
    {
        std::vector<int32_t> cur(pointFaces.off.begin(), pointFaces.off.end() - 1);
        for (int32_t f = 0; f < nF; ++f) {
            const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
            for (int32_t i = 0; i < n; ++i) {
                const int32_t p = facePts[b + i];
                const int32_t slot = cur[p]++;
                // a point listed twice in one face: the reference takes the first occurrence
                // (getNeighbourPoints returns at the first match); keep both entries like pointFaces does
                pointFaces.val[slot] = f;
                pfPrev[slot] = facePts[b + (i == 0 ? n - 1 : i - 1)];
                pfNext[slot] = facePts[b + (i == n - 1 ? 0 : i + 1)];
            }
        }
    }

    lap("pointFaces");
    // ---- pointCells: cells of the point's faces, ascending, unique --------------------------
    // (point ranges on host threads: every range builds its rows as the serial loop would, the ranges are concatenated in order)
    pointCells.off.assign(nP + 1, 0);
    {
        const int parts = rangeParts(nP);
        std::vector<std::vector<int32_t>> pv((size_t)parts);
        std::vector<int32_t> pmax((size_t)parts, 0);
        parallelRanges(nP, parts, [&](int part, int64_t pb, int64_t pe) {
            std::vector<int32_t> tmp;
            std::vector<int32_t> vals;                                  // (thread-local until the end: no shared cache lines in the loop)
            vals.reserve((size_t)(pointFaces.off[pe] - pointFaces.off[pb]));
            int32_t localMax = 0;
            for (int32_t p = (int32_t)pb; p < (int32_t)pe; ++p) {
                tmp.clear();
                for (int32_t k = pointFaces.off[p]; k < pointFaces.off[p + 1]; ++k) {
                    const int32_t f = pointFaces.val[k];
                    tmp.push_back(own[f]);
                    if (f < nIF) tmp.push_back(nei[f]);
                }
                std::sort(tmp.begin(), tmp.end());
                tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                localMax = std::max(localMax, (int32_t)tmp.size());
                vals.insert(vals.end(), tmp.begin(), tmp.end());
                pointCells.off[p + 1] = (int32_t)tmp.size();
            }
            pmax[(size_t)part] = localMax;
            pv[(size_t)part].swap(vals);
        });
        for (int32_t v : pmax) maxPointCells = std::max(maxPointCells, v);
        scanCounts(pointCells.off);
        concatParts(pointCells.val, pv);
    }

    lap("pointCells");
    // ---- edges: bucket (lo -> hi list), sort + unique per bucket => upper-triangular order ----
    // (bucket counts and fills with relaxed atomics on face ranges: the order inside a bucket does not matter, it is sorted next)
    std::vector<int32_t> loOff(nP + 1, 0);
    {
        const int parts = rangeParts(nF);
        std::vector<std::string> perr((size_t)parts);
        parallelRanges(nF, parts, [&](int part, int64_t fb0, int64_t fe0) {
            for (int32_t f = (int32_t)fb0; f < (int32_t)fe0; ++f) {
                const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
                for (int32_t i = 0; i < n; ++i) {
                    const int32_t a = facePts[b + i], c = facePts[b + (i == n - 1 ? 0 : i + 1)];
                    if (a == c) { perr[(size_t)part] = "degenerate edge in face " + std::to_string(f); return; }
                    __atomic_fetch_add(&loOff[(size_t)std::min(a, c) + 1], 1, __ATOMIC_RELAXED);
                }
            }
        });
        for (const std::string& e : perr) if (!e.empty()) return e;
    }
    scanCounts(loOff);
    std::vector<int32_t> his(nnzFP);
    {
        std::vector<int32_t> cur(loOff.begin(), loOff.end() - 1);
        parallelRanges(nF, rangeParts(nF), [&](int, int64_t fb0, int64_t fe0) {
            for (int32_t f = (int32_t)fb0; f < (int32_t)fe0; ++f) {
                const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
                for (int32_t i = 0; i < n; ++i) {
                    const int32_t a = facePts[b + i], c = facePts[b + (i == n - 1 ? 0 : i + 1)];
                    his[(size_t)__atomic_fetch_add(&cur[(size_t)std::min(a, c)], 1, __ATOMIC_RELAXED)] = std::max(a, c);
                }
            }
        });
    }
    std::vector<int32_t> edgeStart(nP + 1, 0);  // first edge id with start == p
    {
        const int parts = rangeParts(nP);
        std::vector<std::vector<int32_t>> pe2((size_t)parts);
        parallelRanges(nP, parts, [&](int part, int64_t pb, int64_t pe) {
            std::vector<int32_t> loc;
            loc.reserve(2 * (size_t)(loOff[pe] - loOff[pb]));
            for (int32_t p = (int32_t)pb; p < (int32_t)pe; ++p) {
                edgeStart[p] = (int32_t)(loc.size() / 2);          // within the range; the range's base is added below
                int32_t* b = his.data() + loOff[p];
                int32_t* e = his.data() + loOff[p + 1];
                std::sort(b, e);
                e = std::unique(b, e);
                for (int32_t* it = b; it != e; ++it) { loc.push_back(p); loc.push_back(*it); }
            }
            pe2[(size_t)part].swap(loc);
        });
        int32_t base = 0;
        for (int part = 0; part < parts; ++part) {
            const int64_t pb = (int64_t)nP * part / parts, pe = (int64_t)nP * (part + 1) / parts;
            if (base) for (int64_t p = pb; p < pe; ++p) edgeStart[(size_t)p] += base;
            base += (int32_t)(pe2[(size_t)part].size() / 2);
        }
        concatParts(edges, pe2);
    }
    nEdges = (int32_t)(edges.size() / 2);
    edgeStart[nP] = nEdges;
    { std::vector<int32_t>().swap(his); }
    auto edgeId = [&](int32_t a, int32_t c) -> int32_t {
        const int32_t lo = std::min(a, c), hi = std::max(a, c);
        for (int32_t e = edgeStart[lo]; e < edgeStart[lo + 1]; ++e)
            if (edges[2 * e + 1] == hi) return e;
        return -1;
    };

    lap("edges");
    // ---- pointEdges / pointPoints ----------------------------------------------------------------
    pointEdges.off.assign(nP + 1, 0);
    for (int32_t e = 0; e < nEdges; ++e) { pointEdges.off[edges[2 * e] + 1]++; pointEdges.off[edges[2 * e + 1] + 1]++; }
    scanCounts(pointEdges.off);
    pointEdges.val.resize(pointEdges.off[nP]);
    pointPoints.resize(pointEdges.off[nP]);
    {
        std::vector<int32_t> cur(pointEdges.off.begin(), pointEdges.off.end() - 1);
        for (int32_t e = 0; e < nEdges; ++e) {
            const int32_t a = edges[2 * e], c = edges[2 * e + 1];
            pointEdges.val[cur[a]] = e; pointPoints[cur[a]++] = c;
            pointEdges.val[cur[c]] = e; pointPoints[cur[c]++] = a;
        }
    }
    for (int32_t p = 0; p < nP; ++p) maxPointPoints = std::max(maxPointPoints, pointEdges.off[p + 1] - pointEdges.off[p]);
    lap("pointEdges");
    if (afterPoints) afterPoints();
    // prev/next vertex of every pointFaces entry as a slot of the point's pointPoints row
    pfPrevSlot.assign(pfPrev.size(), 255);
    pfNextSlot.assign(pfNext.size(), 255);
    parallelRanges(nP, rangeParts(nP), [&](int, int64_t pb, int64_t pe) {
        for (int32_t p = (int32_t)pb; p < (int32_t)pe; ++p) {
            const int32_t nb = pointEdges.off[p], nv = pointEdges.off[p + 1] - nb;
            for (int32_t k = pointFaces.off[p]; k < pointFaces.off[p + 1]; ++k)
                for (int32_t j = 0; j < nv && j < 255; ++j) {
                    if (pointPoints[nb + j] == pfPrev[k]) pfPrevSlot[k] = (uint8_t)j;
                    if (pointPoints[nb + j] == pfNext[k]) pfNextSlot[k] = (uint8_t)j;
                }
        }
    });

    lap("pfPrev/NextSlot");
    // ---- edgeFaces (ascending face id) ------------------------------------------------------------
    edgeFaces.off.assign(nEdges + 1, 0);
    std::vector<int32_t> faceEdge(nnzFP);
    parallelRanges(nF, rangeParts(nF), [&](int, int64_t fb0, int64_t fe0) {
        for (int32_t f = (int32_t)fb0; f < (int32_t)fe0; ++f) {
            const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
            for (int32_t i = 0; i < n; ++i) faceEdge[b + i] = edgeId(facePts[b + i], facePts[b + (i == n - 1 ? 0 : i + 1)]);
        }
    });
    parallelRanges(nnzFP, rangeParts(nnzFP), [&](int, int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) __atomic_fetch_add(&edgeFaces.off[(size_t)faceEdge[(size_t)k] + 1], 1, __ATOMIC_RELAXED);
    });
    scanCounts(edgeFaces.off);
    edgeFaces.val.resize(edgeFaces.off[nEdges]);
    {
        // filled from face ranges in any order, then every row sorted: ascending face id, as the serial fill leaves it
        std::vector<int32_t> cur(edgeFaces.off.begin(), edgeFaces.off.end() - 1);
        parallelRanges(nF, rangeParts(nF), [&](int, int64_t fb0, int64_t fe0) {
            for (int32_t f = (int32_t)fb0; f < (int32_t)fe0; ++f)
                for (int32_t k = faceOffsets[f]; k < faceOffsets[f + 1]; ++k)
                    edgeFaces.val[(size_t)__atomic_fetch_add(&cur[(size_t)faceEdge[(size_t)k]], 1, __ATOMIC_RELAXED)] = f;
        });
        parallelRanges(nEdges, rangeParts(nEdges), [&](int, int64_t e0, int64_t e1) {
            for (int64_t e = e0; e < e1; ++e) std::sort(edgeFaces.val.begin() + edgeFaces.off[(size_t)e], edgeFaces.val.begin() + edgeFaces.off[(size_t)e + 1]);
        });
    }
    { std::vector<int32_t>().swap(faceEdge); }

    lap("edgeFaces");
    // ---- edgeCells (first appearance through edgeFaces, owner then neighbour) + face pairs ----
    edgeCells.off.assign(nEdges + 1, 0);
    {
        const int parts = rangeParts(nEdges);
        std::vector<std::vector<int32_t>> pc((size_t)parts);
        std::vector<std::vector<uint8_t>> p0((size_t)parts), p1((size_t)parts);
        std::vector<std::string> perr((size_t)parts);
        std::vector<int32_t> pmax((size_t)parts, 0);
        parallelRanges(nEdges, parts, [&](int part, int64_t eb0, int64_t ee0) {
            std::vector<int32_t> cells;
            std::vector<int32_t> vals;                                  // (thread-local until the end: no shared cache lines in the loop)
            std::vector<uint8_t> v0, v1;
            const size_t guess = (size_t)(edgeFaces.off[ee0] - edgeFaces.off[eb0]);
            vals.reserve(guess); v0.reserve(guess); v1.reserve(guess);
            int32_t localMax = 0;
            for (int32_t e = (int32_t)eb0; e < (int32_t)ee0; ++e) {
                const int32_t b = edgeFaces.off[e], n = edgeFaces.off[e + 1] - b;
                localMax = std::max(localMax, n);
                if (n > 255) { perr[(size_t)part] = "edge with more than 255 faces"; return; }
                cells.clear();
                for (int32_t i = 0; i < n; ++i) {
                    const int32_t f = edgeFaces.val[b + i];
                    if (std::find(cells.begin(), cells.end(), own[f]) == cells.end()) cells.push_back(own[f]);
                    if (f < nIF && std::find(cells.begin(), cells.end(), nei[f]) == cells.end()) cells.push_back(nei[f]);
                }
                for (int32_t c : cells) {
                    int32_t f0 = -1, f1 = -1, hits = 0;
                    for (int32_t i = 0; i < n; ++i) {
                        const int32_t f = edgeFaces.val[b + i];
                        if (own[f] == c || (f < nIF && nei[f] == c)) {
                            if (hits == 0) f0 = i; else if (hits == 1) f1 = i;
                            ++hits;
                        }
                    }
                    if (hits > 2) { perr[(size_t)part] = "Sanity broken, more than two edge faces belong to same cell"; return; }  // SM.C:1073
                    if (hits < 2) { perr[(size_t)part] = "Sanity broken, didn't find face pairs for cell " + std::to_string(c); return; }  // SM.C:1087
                    vals.push_back(c);
                    v0.push_back((uint8_t)f0);
                    v1.push_back((uint8_t)f1);
                }
                edgeCells.off[e + 1] = (int32_t)cells.size();
            }
            pmax[(size_t)part] = localMax;
            pc[(size_t)part].swap(vals); p0[(size_t)part].swap(v0); p1[(size_t)part].swap(v1);
        });
        for (const std::string& pe : perr) if (!pe.empty()) return pe;   // (the first range in edge order that failed)
        for (int32_t v : pmax) maxEdgeFaces = std::max(maxEdgeFaces, v);
        scanCounts(edgeCells.off);
        concatParts(edgeCells.val, pc);
        concatParts(ecFace0, p0);
        concatParts(ecFace1, p1);
    }

    lap("edgeCells");
    // ---- ring order around each edge -------------------------------------------------------------
    ringFace.assign(edgeFaces.val.size(), -1);
    ringCell.assign(edgeCells.val.size(), -1);
    edgeRingOk.assign((size_t)nEdges, 0);
    parallelRanges(nEdges, rangeParts(nEdges), [&](int, int64_t eb0, int64_t ee0) {
        std::vector<int32_t> deg, usedC;
        for (int32_t e = (int32_t)eb0; e < (int32_t)ee0; ++e) {
            const int32_t fb = edgeFaces.off[e], nf = edgeFaces.off[e + 1] - fb;
            const int32_t cb = edgeCells.off[e], nc = edgeCells.off[e + 1] - cb;
            // a chain has nc = nf - 1 cells, a closed ring nc = nf
            if (nc < 1 || (nc != nf && nc != nf - 1)) continue;
            deg.assign((size_t)nf, 0);
            for (int32_t i = 0; i < nc; ++i) { deg[ecFace0[cb + i]]++; deg[ecFace1[cb + i]]++; }
            bool ok = true;
            int32_t start = 0, nEnds = 0;
            for (int32_t i = 0; i < nf; ++i) {
                if (deg[i] == 1) { if (nEnds == 0) start = i; ++nEnds; }
                else if (deg[i] != 2) ok = false;
            }
            if (!ok || (nc == nf && nEnds != 0) || (nc == nf - 1 && nEnds != 2)) continue;
            usedC.assign((size_t)nc, 0);
            int32_t cur = start, placed = 0;
            ringFace[fb] = edgeFaces.val[fb + cur];
            while (placed < nc) {
                int32_t next = -1, ci = -1;
                for (int32_t i = 0; i < nc; ++i) {
                    if (usedC[i]) continue;
                    if (ecFace0[cb + i] == cur) { next = ecFace1[cb + i]; ci = i; break; }
                    if (ecFace1[cb + i] == cur) { next = ecFace0[cb + i]; ci = i; break; }
                }
                if (ci < 0) break;
                usedC[ci] = 1;
                ringCell[cb + placed] = edgeCells.val[cb + ci];
                ++placed;
                if (placed < nf) ringFace[fb + placed] = edgeFaces.val[fb + next];
                cur = next;
            }
            if (placed == nc && (nc == nf - 1 || cur == start)) edgeRingOk[(size_t)e] = 1;
        }
    });
    lap("rings");
    return "";
}

}  // namespace smgpu
