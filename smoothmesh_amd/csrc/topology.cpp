// topology.cpp -- see topology.hpp.  Count / scan / fill CSR construction, O(nnz).
#include "topology.hpp"

#include <algorithm>
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
                            const int32_t* facePts, const int32_t* own, const int32_t* nei) {
    nPoints = nP; nCells = nC; nFaces = nF; nInternalFaces = nIF;
    if (nP <= 0 || nC <= 0 || nF <= 0 || nIF < 0 || nIF > nF) return "invalid mesh sizes";
    const int64_t nnzFP = faceOffsets[nF];
    if (nnzFP >= (int64_t)1 << 31) return "face-point list exceeds int32 addressing";
    facePoints.off.assign(faceOffsets, faceOffsets + nF + 1);
    facePoints.val.assign(facePts, facePts + nnzFP);
    owner.assign(own, own + nF);
    neighbour.assign(nei, nei + nIF);
    for (int32_t f = 0; f < nF; ++f) {
        const int32_t n = faceOffsets[f + 1] - faceOffsets[f];
        if (n < 3) return "face " + std::to_string(f) + " has fewer than 3 points";
        maxFaceSize = std::max(maxFaceSize, n);
        if (own[f] < 0 || own[f] >= nC) return "owner out of range at face " + std::to_string(f);
        if (f < nIF && (nei[f] < 0 || nei[f] >= nC)) return "neighbour out of range at face " + std::to_string(f);
    }
    for (int64_t i = 0; i < nnzFP; ++i)
        if (facePts[i] < 0 || facePts[i] >= nP) return "face point label out of range";

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

    // ---- pointCells: cells of the point's faces, ascending, unique --------------------------
    pointCells.off.assign(nP + 1, 0);
    {
        std::vector<int32_t> tmp;
        std::vector<int32_t> vals;
        vals.reserve((size_t)nnzFP);
        for (int32_t p = 0; p < nP; ++p) {
            tmp.clear();
            for (int32_t k = pointFaces.off[p]; k < pointFaces.off[p + 1]; ++k) {
                const int32_t f = pointFaces.val[k];
                tmp.push_back(own[f]);
                if (f < nIF) tmp.push_back(nei[f]);
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            maxPointCells = std::max(maxPointCells, (int32_t)tmp.size());
            vals.insert(vals.end(), tmp.begin(), tmp.end());
            pointCells.off[p + 1] = (int32_t)vals.size();
        }
        pointCells.val.swap(vals);
    }

    // ---- edges: bucket (lo -> hi list), sort + unique per bucket => upper-triangular order ----
    std::vector<int32_t> loOff(nP + 1, 0);
    for (int32_t f = 0; f < nF; ++f) {
        const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t a = facePts[b + i], c = facePts[b + (i == n - 1 ? 0 : i + 1)];
            if (a == c) return "degenerate edge in face " + std::to_string(f);
            loOff[std::min(a, c) + 1]++;
        }
    }
    scanCounts(loOff);
    std::vector<int32_t> his(nnzFP);
    {
        std::vector<int32_t> cur(loOff.begin(), loOff.end() - 1);
        for (int32_t f = 0; f < nF; ++f) {
            const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
            for (int32_t i = 0; i < n; ++i) {
                const int32_t a = facePts[b + i], c = facePts[b + (i == n - 1 ? 0 : i + 1)];
                his[cur[std::min(a, c)]++] = std::max(a, c);
            }
        }
    }
    std::vector<int32_t> edgeStart(nP + 1, 0);  // first edge id with start == p
    edges.clear();
    edges.reserve((size_t)nnzFP);
    for (int32_t p = 0; p < nP; ++p) {
        edgeStart[p] = (int32_t)(edges.size() / 2);
        int32_t* b = his.data() + loOff[p];
        int32_t* e = his.data() + loOff[p + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        for (int32_t* it = b; it != e; ++it) { edges.push_back(p); edges.push_back(*it); }
    }
    nEdges = (int32_t)(edges.size() / 2);
    edgeStart[nP] = nEdges;
    edges.shrink_to_fit();
    { std::vector<int32_t>().swap(his); }
    auto edgeId = [&](int32_t a, int32_t c) -> int32_t {
        const int32_t lo = std::min(a, c), hi = std::max(a, c);
        for (int32_t e = edgeStart[lo]; e < edgeStart[lo + 1]; ++e)
            if (edges[2 * e + 1] == hi) return e;
        return -1;
    };

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
    // prev/next vertex of every pointFaces entry as a slot of the point's pointPoints row
    pfPrevSlot.assign(pfPrev.size(), 255);
    pfNextSlot.assign(pfNext.size(), 255);
    for (int32_t p = 0; p < nP; ++p) {
        const int32_t nb = pointEdges.off[p], nv = pointEdges.off[p + 1] - nb;
        for (int32_t k = pointFaces.off[p]; k < pointFaces.off[p + 1]; ++k)
            for (int32_t j = 0; j < nv && j < 255; ++j) {
                if (pointPoints[nb + j] == pfPrev[k]) pfPrevSlot[k] = (uint8_t)j;
                if (pointPoints[nb + j] == pfNext[k]) pfNextSlot[k] = (uint8_t)j;
            }
    }

    // ---- edgeFaces (ascending face id) ------------------------------------------------------------
    edgeFaces.off.assign(nEdges + 1, 0);
    std::vector<int32_t> faceEdge(nnzFP);
    for (int32_t f = 0; f < nF; ++f) {
        const int32_t b = faceOffsets[f], n = faceOffsets[f + 1] - b;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t e = edgeId(facePts[b + i], facePts[b + (i == n - 1 ? 0 : i + 1)]);
            faceEdge[b + i] = e;
            edgeFaces.off[e + 1]++;
        }
    }
    scanCounts(edgeFaces.off);
    edgeFaces.val.resize(edgeFaces.off[nEdges]);
    {
        std::vector<int32_t> cur(edgeFaces.off.begin(), edgeFaces.off.end() - 1);
        for (int32_t f = 0; f < nF; ++f)
            for (int32_t k = faceOffsets[f]; k < faceOffsets[f + 1]; ++k) edgeFaces.val[cur[faceEdge[k]]++] = f;
    }
    { std::vector<int32_t>().swap(faceEdge); }

    // ---- edgeCells (first appearance through edgeFaces, owner then neighbour) + face pairs ----
    edgeCells.off.assign(nEdges + 1, 0);
    edgeCells.val.reserve(edgeFaces.val.size());
    ecFace0.reserve(edgeFaces.val.size());
    ecFace1.reserve(edgeFaces.val.size());
    {
        std::vector<int32_t> cells;
        for (int32_t e = 0; e < nEdges; ++e) {
            const int32_t b = edgeFaces.off[e], n = edgeFaces.off[e + 1] - b;
            maxEdgeFaces = std::max(maxEdgeFaces, n);
            if (n > 255) return "edge with more than 255 faces";
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
                if (hits > 2) return "Sanity broken, more than two edge faces belong to same cell";  // SM.C:1073
                if (hits < 2) return "Sanity broken, didn't find face pairs for cell " + std::to_string(c);  // SM.C:1087
                edgeCells.val.push_back(c);
                ecFace0.push_back((uint8_t)f0);
                ecFace1.push_back((uint8_t)f1);
            }
            edgeCells.off[e + 1] = (int32_t)edgeCells.val.size();
        }
    }

    // ---- ring order around each edge -------------------------------------------------------------
    ringFace.assign(edgeFaces.val.size(), -1);
    ringCell.assign(edgeCells.val.size(), -1);
    edgeRingOk.assign((size_t)nEdges, 0);
    {
        std::vector<int32_t> deg, usedC;
        for (int32_t e = 0; e < nEdges; ++e) {
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
    }
    return "";
}

}  // namespace smgpu
