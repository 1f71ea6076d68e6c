// tiles.cpp -- see tiles.hpp.
#include "tiles.hpp"

#include <algorithm>

namespace smgpu {

std::string GeomTiles::build(const Topology& t, int32_t capCells, int32_t capPoints, int32_t capFaces) {
    const auto& cf = t.cellFacesGeom;
    const auto& fp = t.facePoints;
    std::vector<int32_t> stampP((size_t)t.nPoints, -1), stampF((size_t)t.nFaces, -1);
    // pass 1: greedy tile boundaries under the three capacities
    cellBeg.assign(1, 0);
    int32_t tile = 0, nP = 0, nF = 0, nC = 0;
    for (int32_t c = 0; c < t.nCells; ++c) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            int32_t addF = 0, addP = 0;
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t f = cf.val[k] & 0x7fffffff;
                if (stampF[f] != tile) { stampF[f] = tile; ++addF; }
                for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j)
                    if (stampP[fp.val[j]] != tile) { stampP[fp.val[j]] = tile; ++addP; }
            }
            if (nC > 0 && (nC + 1 > capCells || nP + addP > capPoints || nF + addF > capFaces)) {
                cellBeg.push_back(c);   // close the tile before this cell and re-add the cell to a fresh one
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

    // pass 2: per tile unique lists (ascending) and local indices
    std::fill(stampP.begin(), stampP.end(), -1);
    std::fill(stampF.begin(), stampF.end(), -1);
    std::vector<int32_t> locP((size_t)t.nPoints, -1), locF((size_t)t.nFaces, -1);
    tpOff.assign(1, 0); tfOff.assign(1, 0); tfpOff.assign(1, 0);
    tpIds.clear(); tfIds.clear(); tfpLoc.clear();
    cfLoc.assign(cf.val.size(), 0);
    std::vector<int32_t> faces, points;
    for (int32_t ti = 0; ti < nTiles; ++ti) {
        faces.clear(); points.clear();
        const int32_t cb = cellBeg[ti], ce = cellBeg[ti + 1];
        for (int32_t c = cb; c < ce; ++c)
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t f = cf.val[k] & 0x7fffffff;
                if (stampF[f] != ti) { stampF[f] = ti; faces.push_back(f); }
            }
        std::sort(faces.begin(), faces.end());
        for (int32_t f : faces)
            for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j)
                if (stampP[fp.val[j]] != ti) { stampP[fp.val[j]] = ti; points.push_back(fp.val[j]); }
        std::sort(points.begin(), points.end());
        if ((int32_t)points.size() > 32767 || (int32_t)faces.size() > 32767) return "tile too large for 15-bit local indices";
        for (size_t i = 0; i < points.size(); ++i) locP[points[i]] = (int32_t)i;
        for (size_t i = 0; i < faces.size(); ++i) locF[faces[i]] = (int32_t)i;
        tpIds.insert(tpIds.end(), points.begin(), points.end());
        tpOff.push_back((int32_t)tpIds.size());
        for (int32_t f : faces) {
            const bool ownerHere = t.owner[f] >= cb && t.owner[f] < ce;
            tfIds.push_back(ownerHere ? (int32_t)(0x80000000u | (uint32_t)f) : f);
            for (int32_t j = fp.off[f]; j < fp.off[f + 1]; ++j) tfpLoc.push_back((uint16_t)locP[fp.val[j]]);
            tfpOff.push_back((int32_t)tfpLoc.size());
        }
        tfOff.push_back((int32_t)tfIds.size());
        for (int32_t c = cb; c < ce; ++c)
            for (int32_t k = cf.off[c]; k < cf.off[c + 1]; ++k) {
                const int32_t v = cf.val[k];
                cfLoc[k] = (uint16_t)(locF[v & 0x7fffffff] | (v < 0 ? 0x8000 : 0));
            }
        maxPoints = std::max(maxPoints, (int32_t)points.size());
        maxFaces = std::max(maxFaces, (int32_t)faces.size());
        maxCells = std::max(maxCells, ce - cb);
    }
    return "";
}

std::string SmoothTiles::build(const Topology& t, int32_t capTile, int32_t capCells, int32_t capPoints) {
    const auto& pc = t.pointCells;
    const auto& pe = t.pointEdges;   // offsets shared with pointPoints
    std::vector<int32_t> stampC((size_t)t.nCells, -1), stampN((size_t)t.nPoints, -1);
    ptBeg.assign(1, 0);
    int32_t tile = 0, nC = 0, nN = 0, nT = 0;
    for (int32_t p = 0; p < t.nPoints; ++p) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            int32_t addC = 0, addN = 0;
            for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k)
                if (stampC[pc.val[k]] != tile) { stampC[pc.val[k]] = tile; ++addC; }
            if (stampN[p] != tile) { stampN[p] = tile; ++addN; }
            for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k)
                if (stampN[t.pointPoints[k]] != tile) { stampN[t.pointPoints[k]] = tile; ++addN; }
            if (nT > 0 && (nT + 1 > capTile || nC + addC > capCells || nN + addN > capPoints)) {
                ptBeg.push_back(p);
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
    pcLoc.assign(pc.val.size(), 0);
    ppLoc.assign(t.pointPoints.size(), 0);
    selfLoc.assign((size_t)t.nPoints, 0);
    std::vector<int32_t> cells, pts;
    for (int32_t ti = 0; ti < nTiles; ++ti) {
        cells.clear(); pts.clear();
        const int32_t pb = ptBeg[ti], pend = ptBeg[ti + 1];
        for (int32_t p = pb; p < pend; ++p) {
            for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k)
                if (stampC[pc.val[k]] != ti) { stampC[pc.val[k]] = ti; cells.push_back(pc.val[k]); }
            if (stampN[p] != ti) { stampN[p] = ti; pts.push_back(p); }
            for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k)
                if (stampN[t.pointPoints[k]] != ti) { stampN[t.pointPoints[k]] = ti; pts.push_back(t.pointPoints[k]); }
        }
        std::sort(cells.begin(), cells.end());
        std::sort(pts.begin(), pts.end());
        if ((int32_t)cells.size() > 32767 || (int32_t)pts.size() > 32767) return "tile too large for 15-bit local indices";
        for (size_t i = 0; i < cells.size(); ++i) locC[cells[i]] = (int32_t)i;
        for (size_t i = 0; i < pts.size(); ++i) locN[pts[i]] = (int32_t)i;
        tcIds.insert(tcIds.end(), cells.begin(), cells.end());
        tcOff.push_back((int32_t)tcIds.size());
        tnIds.insert(tnIds.end(), pts.begin(), pts.end());
        tnOff.push_back((int32_t)tnIds.size());
        for (int32_t p = pb; p < pend; ++p) {
            selfLoc[p] = (uint16_t)locN[p];
            for (int32_t k = pc.off[p]; k < pc.off[p + 1]; ++k) pcLoc[k] = (uint16_t)locC[pc.val[k]];
            for (int32_t k = pe.off[p]; k < pe.off[p + 1]; ++k) ppLoc[k] = (uint16_t)locN[t.pointPoints[k]];
        }
        maxCells = std::max(maxCells, (int32_t)cells.size());
        maxPoints = std::max(maxPoints, (int32_t)pts.size());
        maxTilePoints = std::max(maxTilePoints, pend - pb);
    }

    // pairShare: neighbours i, j of p share a cell  <=>  pointCells(q_i) and pointCells(q_j) intersect
    pairShare.assign(t.pointPoints.size(), 0);
    if (t.maxPointPoints <= 16) {
        for (int32_t p = 0; p < t.nPoints; ++p) {
            const int32_t b = pe.off[p], v = pe.off[p + 1] - b;
            for (int32_t i = 0; i < v; ++i) {
                const int32_t qi = t.pointPoints[b + i];
                for (int32_t j = i + 1; j < v; ++j) {
                    const int32_t qj = t.pointPoints[b + j];
                    int32_t a = pc.off[qi], ae = pc.off[qi + 1], c = pc.off[qj], ce = pc.off[qj + 1];
                    bool share = false;
                    while (a < ae && c < ce) {
                        if (pc.val[a] == pc.val[c]) { share = true; break; }
                        if (pc.val[a] < pc.val[c]) ++a; else ++c;
                    }
                    if (share) { pairShare[b + i] |= (uint16_t)(1u << j); pairShare[b + j] |= (uint16_t)(1u << i); }
                }
            }
        }
    }
    return "";
}

}  // namespace smgpu
