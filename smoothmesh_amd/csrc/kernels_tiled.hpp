// kernels_tiled.hpp -- LDS-staged forms of the two gather stages (see tiles.hpp for the idea).
// Same arithmetic, same summation order, same results as k_face_geom + k_cell_centres and k_smooth in
// kernels.hpp; global memory is only streamed (coalesced id-list loads), all indexed access is in LDS.
#pragma once
#include "kernels.hpp"

namespace smgpu {

struct GeomTileView {
    const int* cellBeg; const int* tpOff; const int* tpIds; const int* tfOff; const int* tfIds;
    const int* tfpOff; const uint16_t* tfpLoc; const uint16_t* cfLoc;
    int maxPoints, maxFaces;
};

struct SmoothTileView {
    const int* ptBeg; const int* tcOff; const int* tcIds; const int* tnOff; const int* tnIds;
    const uint16_t* pcLoc; const uint16_t* ppLoc;   // ppLoc bit 15: the neighbour is an internal point
    const uint16_t* selfLoc; const uint16_t* pairShare;
    int maxCells, maxPoints, usePairShare;
};

__device__ __forceinline__ V3 ldsv(const double* x, const double* y, const double* z, int i) { return v3(x[i], y[i], z[i]); }

// Cell centres of the current coordinates for one tile of consecutive cells:
// OpenFOAM makeFaceCentresAndAreas + makeCellCentresAndVols (.com v2412), staged through LDS.
template <int T>
__global__ void __launch_bounds__(T) k_geom_tile(MeshView m, State s, GeomTileView g, int wantAvg, int writeFaces) {
    if (s.acc->stop) return;
    extern __shared__ double lds[];
    double* px = lds;             double* py = px + g.maxPoints;  double* pz = py + g.maxPoints;
    double* fcx = pz + g.maxPoints; double* fcy = fcx + g.maxFaces; double* fcz = fcy + g.maxFaces;
    double* fax = fcz + g.maxFaces; double* fay = fax + g.maxFaces; double* faz = fay + g.maxFaces;
    const int tile = blockIdx.x, tid = threadIdx.x;

    // phase 0: the tile's points (ascending ids: near-contiguous 24-byte records)
    {
        const int b = g.tpOff[tile], n = g.tpOff[tile + 1] - b;
        for (int i = tid; i < n; i += T) {
            const V3 v = ldv(s.ptsCur, g.tpIds[b + i]);
            px[i] = v.x; py[i] = v.y; pz[i] = v.z;
        }
    }
    __syncthreads();

    // phase 1: every face of the tile once
    {
        const int b = g.tfOff[tile], nf = g.tfOff[tile + 1] - b;
        for (int i = tid; i < nf; i += T) {
            const int o = g.tfpOff[b + i], n = g.tfpOff[b + i + 1] - o;
            const uint16_t* lp = g.tfpLoc + o;
            V3 ctr, area;
            V3 fCentre = ldsv(px, py, pz, lp[0]);
            for (int j = 1; j < n; ++j) fCentre = fCentre + ldsv(px, py, pz, lp[j]);
            fCentre = fCentre / double(n);
            if (n == 3) {
                const V3 p0 = ldsv(px, py, pz, lp[0]), p1 = ldsv(px, py, pz, lp[1]), p2 = ldsv(px, py, pz, lp[2]);
                ctr = (1.0 / 3.0) * ((p0 + p1) + p2);
                area = 0.5 * cross(p1 - p0, p2 - p0);
            } else {
                V3 sumN = v3(0, 0, 0), sumAc = v3(0, 0, 0);
                double sumA = 0.0;
                V3 thisPoint = ldsv(px, py, pz, lp[0]);
                const V3 first = thisPoint;
                for (int j = 0; j < n; ++j) {
                    const V3 nextPoint = (j == n - 1) ? first : ldsv(px, py, pz, lp[j + 1]);
                    const V3 c = (thisPoint + nextPoint) + fCentre;
                    const V3 nn = cross(nextPoint - thisPoint, fCentre - thisPoint);
                    const double a = mag(nn);
                    sumN = sumN + nn;
                    sumA += a;
                    sumAc = sumAc + a * c;
                    thisPoint = nextPoint;
                }
                if (sumA < SMGPU_ROOTVSMALL) { ctr = fCentre; area = v3(0, 0, 0); }
                else { ctr = ((1.0 / 3.0) * sumAc) / sumA; area = 0.5 * sumN; }
            }
            fcx[i] = ctr.x; fcy[i] = ctr.y; fcz[i] = ctr.z;
            fax[i] = area.x; fay[i] = area.y; faz[i] = area.z;
            const int fid = g.tfIds[b + i];
            if (fid < 0) {   // this tile holds the face's owner cell: it publishes the per-face values
                const int f = fid & 0x7fffffff;
                if (wantAvg) stv(s.fAvg, f, fCentre);
                if (writeFaces) { stv(s.fCtr, f, ctr); stv(s.fArea, f, area); }
            }
        }
    }
    __syncthreads();

    // phase 2: one thread per cell
    const int c = g.cellBeg[tile] + tid;
    if (c < g.cellBeg[tile + 1]) {
        const int b = m.cfOff[c], e = m.cfOff[c + 1];
        V3 cEst = v3(0, 0, 0);
        for (int k = b; k < e; ++k) cEst = cEst + ldsv(fcx, fcy, fcz, g.cfLoc[k] & 0x7fff);
        cEst = cEst / double(e - b);
        V3 ctr = v3(0, 0, 0);
        double vol = 0.0;
        for (int k = b; k < e; ++k) {
            const int v = g.cfLoc[k];
            const int f = v & 0x7fff;
            const V3 fc = ldsv(fcx, fcy, fcz, f);
            const V3 fA = ldsv(fax, fay, faz, f);
            const double pyr3Vol = (v & 0x8000) ? dot(fA, cEst - fc) : dot(fA, fc - cEst);
            const V3 pc = (3.0 / 4.0) * fc + (1.0 / 4.0) * cEst;
            ctr = ctr + pyr3Vol * pc;
            vol += pyr3Vol;
        }
        if (fabs(vol) > SMGPU_VSMALL) ctr = ctr / vol;
        else ctr = cEst;
        stv(s.cellCtr, c, ctr);
    }
}

// The fused per-point proposal kernel of kernels.hpp (k_smooth) with the cell centres and neighbour
// coordinates of one tile of consecutive points staged in LDS.
template <bool FINAL, int T>
__global__ void __launch_bounds__(T) k_smooth_tile(MeshView m, State s, Prm prm, SmoothTileView g) {
    if (s.acc->stop) return;
    extern __shared__ double lds[];
    double* cx = lds;              double* cy = cx + g.maxCells;  double* cz = cy + g.maxCells;
    double* nx = cz + g.maxCells;  double* ny = nx + g.maxPoints; double* nz = ny + g.maxPoints;
    const int tile = blockIdx.x, tid = threadIdx.x;
    {
        const int b = g.tcOff[tile], n = g.tcOff[tile + 1] - b;
        for (int i = tid; i < n; i += T) {
            const V3 v = ldv(s.cellCtr, g.tcIds[b + i]);
            cx[i] = v.x; cy[i] = v.y; cz[i] = v.z;
        }
        const int b2 = g.tnOff[tile], n2 = g.tnOff[tile + 1] - b2;
        for (int i = tid; i < n2; i += T) {
            const V3 v = ldv(s.ptsCur, g.tnIds[b2 + i]);
            nx[i] = v.x; ny[i] = v.y; nz[i] = v.z;
        }
    }
    __syncthreads();

    const int p = g.ptBeg[tile] + tid;
    double dist = 0.0;
    int fcount = 0;
    if (p < g.ptBeg[tile + 1]) {
        const uint8_t fl = m.pflags[p];
        const bool internal = fl & PF_INTERNAL;
        const V3 cur = ldsv(nx, ny, nz, g.selfLoc[p]);
        const int nb = m.ppOff[p], ne = m.ppOff[p + 1];
        V3 sum = v3(0, 0, 0), r1, r2, r3;
        int count = 0, hc = 0;
        const int slot = s.sharedSlot ? s.sharedSlot[p] : -1;
        if (slot >= 0) {
            const double* r = s.combA + (size_t)slot * SMGPU_HALO_A_DOUBLES;
            sum = v3(r[0], r[1], r[2]);
            r1 = v3(r[3], r[4], r[5]); r2 = v3(r[6], r[7], r[8]); r3 = v3(r[9], r[10], r[11]);
            const long long pk = __double_as_longlong(r[12]);
            count = (int)(pk & 0xffffffffll);
            hc = (int)(pk >> 32);
        } else {
            if (internal) {   // SM.C:116-130
                const int b = m.pcOff[p], e = m.pcOff[p + 1];
                count = e - b;
                for (int k = b; k < e; ++k) sum = sum + ldsv(cx, cy, cz, g.pcLoc[k]);
            }
            // SM.C:325-387 (stable top three; boundary points look at boundary neighbours only)
            double l1 = 0, l2 = 0, l3 = 0;
            int k1 = -1, k2 = -1, k3 = -1;
            for (int k = nb; k < ne; ++k) {
                const int q = g.ppLoc[k];
                if (!internal && (q & 0x8000)) continue;
                const double len = mag(cur - ldsv(nx, ny, nz, q & 0x7fff));
                if (k1 < 0 || len < l1) { l3 = l2; k3 = k2; l2 = l1; k2 = k1; l1 = len; k1 = k; }
                else if (k2 < 0 || len < l2) { l3 = l2; k3 = k2; l2 = len; k2 = k; }
                else if (k3 < 0 || len < l3) { l3 = len; k3 = k; }
            }
            if (k2 < 0) { s.acc->err = 1; r1 = r2 = r3 = v3(0, 0, 0); }
            else {
                r1 = ldsv(nx, ny, nz, g.ppLoc[k1] & 0x7fff) - cur;
                r2 = ldsv(nx, ny, nz, g.ppLoc[k2] & 0x7fff) - cur;
                r3 = (k3 < 0) ? v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT) : ldsv(nx, ny, nz, g.ppLoc[k3] & 0x7fff) - cur;
                if (g.usePairShare) hc = (g.pairShare[k1] >> (k2 - nb)) & 1;
                else hc = shareCell(m, m.ppPt[k1], m.ppPt[k2]) ? 1 : 0;
            }
        }
        V3 np = cur;
        if (count) np = sum / double(count);                          // SM.C:155-163
        const double blendFrac = arRatio(r1, r2, r3, hc != 0, internal);
        if (blendFrac > 0.0) {                                         // SM.C:580-590
            const V3 aCoords = cur + (r1 + r2) / 2.0;
            np = (1.0 - blendFrac) * np + blendFrac * aCoords;
        }
        {                                                              // SM.C:722-745
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        bool frozen = false;                                           // SM.C:611-648
        {
            double shortestCur = SMGPU_GREAT, shortestNew = SMGPU_GREAT;
            for (int k = nb; k < ne; ++k) {
                const V3 q = ldsv(nx, ny, nz, g.ppLoc[k] & 0x7fff);
                const double tc = mag(cur - q);
                if (tc < shortestCur) shortestCur = tc;
                const double tn = mag(np - q);
                if (tn < shortestNew) shortestNew = tn;
            }
            const double shortest = (shortestNew < shortestCur) ? shortestNew : shortestCur;
            if (prm.totalMinFreeze && (shortest < prm.minEdge)) frozen = true;
            else if ((shortestNew < prm.minEdge) && (shortestNew < shortestCur)) frozen = true;
        }
        if (FINAL) {
            if (frozen || (!internal && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
            dist = mag(np - cur) / prm.maxStep;
            stv(s.ptsNext, p, np);
        } else {
            stv(s.prop, p, np);
            s.frozen[p] = frozen ? 1 : 0;
        }
    }
    if (FINAL) {
        // block reduction as blockAccumulate, for T threads
        __shared__ double shMax[T / 64];
        __shared__ int shCnt[T / 64];
        if (!(dist > 0.0)) dist = 0.0;
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_down(dist, o, 64);
            const int oc = __shfl_down(fcount, o, 64);
            dist = (od > dist) ? od : dist;
            fcount += oc;
        }
        if ((tid & 63) == 0) { shMax[tid >> 6] = dist; shCnt[tid >> 6] = fcount; }
        __syncthreads();
        if (tid == 0) {
            double d = shMax[0]; int c = shCnt[0];
            for (int i = 1; i < T / 64; ++i) { d = (shMax[i] > d) ? shMax[i] : d; c += shCnt[i]; }
            if (d > 0.0) atomicMax(&s.acc->resBits, (unsigned long long)__double_as_longlong(d));
            if (c) atomicAdd(&s.acc->nFrozen, c);
        }
    }
}

}  // namespace smgpu
