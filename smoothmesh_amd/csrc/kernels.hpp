// kernels.hpp -- HIP kernels (gfx950, wave64) for the smoothing iteration.
//
// Compiled with -ffp-contract=off: every expression keeps the reference's evaluation order so
// that coordinates are bit-comparable with the x86-64 reference arithmetic (sqrt and division
// are correctly rounded on both; only acos, which feeds threshold comparisons only, differs in
// its last bits between glibc and ROCm's ocml).
//
// Reference lines restated by each kernel are cited at the kernel (SM.C = src/smoothMesh.C).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smgpu.h"
#include "vec3.hpp"
#include "smacos.hpp"

namespace smgpu {

constexpr int kBlock = 256;

// point flag bits
constexpr uint8_t PF_INTERNAL = 1;   // isInternalPoint, SM.C:40-91
constexpr uint8_t PF_SMOOTHSURF = 2; // isSmoothingSurfacePoint, BPS.C:404-412

struct MeshView {
    int nPoints, nCells, nFaces, nInternalFaces, nEdges;
    const int* faceOff; const int* facePts;
    const int* cfOff; const int* cfVal;          // cell -> faces, geometry order, bit31 = neighbour side
    const int* pcOff; const int* pcVal;          // pointCells
    const int* ppOff; const int* ppPt; const int* peEdge;  // pointPoints / pointEdges (shared offsets)
    const int* pfOff; const int* pfPrev; const int* pfNext;  // pointFaces entries -> prev/next vertex
    const int* pfFace;                                        // pointFaces (face ids, ascending)
    const uint8_t* pfPrevSlot; const uint8_t* pfNextSlot;    // ... as slots of the point's pointPoints row
    const int* ringFace; const int* ringCell; const uint8_t* edgeRingOk;  // ring order around each edge
    const int* edges;                             // 2 per edge
    const int* efOff; const int* efFace;          // edgeFaces
    const int* ecOff; const int* ecCell; const uint8_t* ecF0; const uint8_t* ecF1;  // edgeCells + face pair
    const uint8_t* pflags;
};

struct Accum {                 // device-side loop state
    int nActive;               // face-angle walk: points outside the good range (this iteration)
    int nEaMaybe, nFaMaybe;    // elements the f32 filters could not decide (this iteration)
    int stop;                  // set once residual < relTol (SM.C:2401)
    int err;                   // 1 = fewer than two closest points (SM.C:354-362), 2 = too many sharing ranks, 3 = walk barrier timed out
    int nFaPts, nFaEdges;      // face-angle pass: points / edges listed for the exact evaluation (this iteration; adjacent: one scan writes both)
    int nNearTies;             // angle comparisons of this iteration whose two sides were 1 .. Prm::nearUlps ulp apart (noteNear below)
};

// Peer-store transport of the shared-point records (multi-rank; smgpu_halo_set_push).  Every rank maps its peers' receive
// buffers and flag words (hipIpc).  The kernels that produce exchange A / L / F then store each record where it is consumed --
// the slot of the PEER's receive buffer (slotA / slotL / slotF: one address per send slot) -- and the last workgroup of the
// producing launch to finish raises this rank's flag at every peer (pushSignal); the first kernel that consumes the records
// waits for its peers' flags (pushWait).  pack = send: no collective kernel, no host call, nothing between pack and combine
// but the wire.  Flags carry the iteration number (monotone), one word per (peer, exchange kind).
struct PushView {
    double* const* slotA; double* const* slotL; int* const* slotF;   // [nSend] destinations; NULL = the local send buffers
    unsigned* ticket;              // [2] arrivals of the producing launch's workgroups (kind 0 = A + L, 1 = F)
    unsigned* const* peerFlag;     // [nPeers] this rank's pair of flag words in peer o's flag array
    const unsigned* localFlag;     // [2 * nPeers] the flag words the peers write here: [2 * o + kind]
    int nPeers;
    int fence;                     // SMGPU_PUSH_FENCE=1: full system-scope release / acquire fences around the hand-off (A/B; see pushSignal)
};
// A record word on its way to a peer: a system-scope store (sc0 sc1: written through to the peer's memory, nothing left dirty in
// this GPU's L2).  The receive buffers are uncached allocations (smgpu_push_alloc), so neither side needs a cache write-back
// or invalidate for the records -- a system-scope release fence here is a write-back of EVERYTHING dirty in the L2 (the cell
// centres the geometry kernel has just written: ~10 us on the critical path), an acquire on the consuming side an invalidate
// of everything the next kernels are about to re-read.
__device__ __forceinline__ void stPeer(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void stPeer(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// two record words in ONE 16-byte system-scope store (p 16-byte aligned): a write-through store is one fabric write whatever its
// width, so 8-byte stores cost 2.7x the time per byte of 16-byte ones (MI355X_MICROARCH.md, "stores of each flavour").  The
// waves drain them with s_waitcnt vmcnt(0) in pushSignal (the compiler does not count inline-assembly stores).
__device__ __forceinline__ void stPeer2(double* p, double a, double b) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    u32x4 v;
    v.x = (unsigned)ba; v.y = (unsigned)(ba >> 32); v.z = (unsigned)bb; v.w = (unsigned)(bb >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
constexpr int PUSH_ERR_TIMEOUT = 6;   // Accum::err: a peer's records did not arrive (pushWait)
// what a consuming kernel needs to wait for the records of this iteration (localFlag == NULL: nothing to wait for)
struct PushWait { const unsigned* localFlag; int nPeers; int kind; unsigned tag; int* err; int fence; unsigned long long timeoutTicks; };

// End of a producing kernel, called by EVERY thread of EVERY workgroup of the launch: when the last workgroup has stored its
// records, this rank's flag goes up at every peer.  Order: every wave drains its (write-through) stores, the workgroup meets,
// one lane takes a ticket; the workgroup that draws the last ticket resets the counter for the next launch and stores the
// flags (system scope).
// nBlocks: the workgroups of the launch that call this (default: all of them; a launch whose workgroups play several roles
// passes the size of the producing role)
__device__ __forceinline__ void pushSignal(const PushView& pv, int kind, unsigned tag, unsigned nBlocks = 0) {
    if (!pv.ticket) return;
    if (nBlocks == 0) nBlocks = gridDim.x;
    __shared__ int lastWg;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (pv.fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned t = __hip_atomic_fetch_add(&pv.ticket[kind], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lastWg = (t == nBlocks - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!lastWg) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&pv.ticket[kind], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)threadIdx.x < pv.nPeers) {
        if (pv.fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_store(pv.peerFlag[threadIdx.x] + kind, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Start of the first consuming kernel, called by every thread: lane o waits for peer o's flag (a bounded wait: timeoutTicks of
// the 100 MHz wall clock -- SMGPU_PUSH_TIMEOUT_S, default 60 s: ranks may drift apart by seconds on the host, e.g. while one of
// them still writes its sub-domain -- then Accum::err; once that error is up no later wait spins again: the run is lost, the
// host finds the word at the end of the chunk), acquires at system scope, the workgroup meets; plain loads of the records follow.
__device__ __forceinline__ void pushWait(const PushWait& pw) {
    if (!pw.localFlag) return;
    if ((int)threadIdx.x < pw.nPeers) {
        const unsigned* f = pw.localFlag + 2 * threadIdx.x + pw.kind;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        const bool lost = __hip_atomic_load(pw.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PUSH_ERR_TIMEOUT;
        while (!lost && (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - pw.tag) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > pw.timeoutTicks) { *pw.err = PUSH_ERR_TIMEOUT; break; }
        }
        if (pw.fence) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
}

// ---- dependencies BETWEEN the workgroups of one launch (kernels whose workgroups play several roles, kernels_tiled.hpp) ----
// A record written by one workgroup and read by another of the same launch, possibly on another XCD (each XCD has its own L2,
// and nothing snoops between them inside a kernel): stored and loaded with agent-scope accesses (sc1: the store goes through to
// memory, the load does not trust a line of the local L2), 8 bytes at a time.  An agent-scope release FENCE would do for plain
// stores, but it is a write-back of everything dirty in the XCD's L2, per producing workgroup.
__device__ __forceinline__ void stCoh(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ldCoh(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// a record another KERNEL has written while this one was running (the host's exchange on its own stream, ordered by flag words
// instead of kernel boundaries -- smgpu.hip, the "flagged" arrangement): system-scope loads, past both cache levels
__device__ __forceinline__ double ldSys(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ int ldSys(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ V3 ldvCoh(const double* base, int i) {
    const double* p = base + 3 * (size_t)i;
    return v3(ldCoh(p), ldCoh(p + 1), ldCoh(p + 2));
}
__device__ __forceinline__ void stvCoh(double* base, int i, const V3& v) {
    double* p = base + 3 * (size_t)i;
    stCoh(p, v.x); stCoh(p + 1, v.y); stCoh(p + 2, v.z);
}
// "the producing role's workgroups are done".  Counters and flag are monotone (no reset between launches: targets are multiples of
// the launch's serial number).  The producers COUNT -- on a counter per XCD (eight words on eight cache lines: a returning atomic
// on one word runs at ~90 per microsecond chip-wide, 900 arrivals on one word are 10 us), the last arrival of an XCD on a second
// level, the last of those raises the FLAG word -- and the consumers poll the flag with plain agent-scope loads.  (First version:
// consumers polling the counter itself.  275 polling workgroups and 900 arrivals on one word: the arrivals queued behind the
// polls, the count reached its target ~30 us late.)
// Producer side: called by every thread of the workgroup after its coherent stores; consumer side: by every thread before its
// coherent loads (bounded wait: a role order the hardware does not dispatch in -- or a device shared with other engines' spinning
// launches -- raises Accum::err instead of hanging).
constexpr int ROLE_ERR_TIMEOUT = 7;   // Accum::err: a role of a multi-role launch waited for an earlier role for too long
constexpr int kRoleStride = 32;       // unsigned words between two counters (128 bytes)
constexpr int kRoleWords = 10 * kRoleStride;   // [0..7] per-XCD arrivals, [8] XCDs done, [9] flag
// nPerXcd: producing workgroups per XCD (the role's workgroup count is a multiple of 8 and starts at a multiple of 8)
__device__ __forceinline__ void roleDone(unsigned* tk, unsigned serial, unsigned nPerXcd) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = __hip_atomic_fetch_add(tk + (blockIdx.x & 7u) * kRoleStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1u == serial * nPerXcd) {
            const unsigned b = __hip_atomic_fetch_add(tk + 8 * kRoleStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1u == serial * 8u) __hip_atomic_store(tk + 9 * kRoleStride, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// a role of a few workgroups anywhere in the grid: one counter ([8]), then the flag
__device__ __forceinline__ void roleDoneSmall(unsigned* tk, unsigned serial, unsigned nBlocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b = __hip_atomic_fetch_add(tk + 8 * kRoleStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b + 1u == serial * nBlocks) __hip_atomic_store(tk + 9 * kRoleStride, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void roleWait(const unsigned* tk, unsigned serial, int* err) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(tk + 9 * kRoleStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - serial) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if ((++spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *err = ROLE_ERR_TIMEOUT; break; }   // 2 s
        }
    }
    __syncthreads();
}

// One wave on the host's exchange stream (the flagged arrangement, smgpu.hip): raise a word behind the exchange that was enqueued
// in front of this kernel (the role that consumes the records polls it) and / or wait for the word the role that packs the NEXT
// exchange raises -- what hipStreamWriteValue32 + hipStreamWaitValue32 do as two kernels with a dispatch gap between them.
__global__ void __launch_bounds__(64) k_flag_relay(unsigned* writeWord, unsigned writeValue, const unsigned* waitWord, unsigned waitValue, int* err,
                                                    unsigned long long timeoutTicks) {
    if (threadIdx.x != 0) return;
    if (writeWord) __hip_atomic_store(writeWord, writeValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!waitWord) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(waitWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - waitValue) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;      // (the run is lost already)
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeoutTicks) { *err = PUSH_ERR_TIMEOUT; break; }
        }
    }
}

struct State {
    double* ptsCur; double* ptsNext; double* prop;
    double* stepSqr;   // != NULL: the proposal kernel leaves |proposal - current|^2 per point for k_apply_swap
    double* fCtr; double* fArea; double* cellCtr;
    double* fAvg;      // face vertex averages for the f32 face-angle filter: by face id, or (avgPacked) in the order of the
                       // geometry tiles' face lists -- contiguous stores; the exact kernels form the average themselves
    int avgPacked;
    uint8_t* frozen;
    double* edgeMin; double* edgeMax; double* ptMin; double* ptMax;
    uint8_t* faActive; uint8_t* faS; uint8_t* faN; int* walkStack;
    int* faEdgeList; int* faPointList;   // exact face-angle pass on lists (k_fa_collect)
    uint8_t faGen;     // generation tag of this iteration's faActive / faMaybe marks (a mark counts only if it equals the tag: no
                       // per-iteration clearing of the two P-byte arrays; they are zeroed when the tag wraps)
    Accum* acc;
    double* blkMax; int* blkCnt;   // per-workgroup partials of (max step, frozen count); reduced by k_finish
    smgpu_iter_stats* stats;
    const int* sharedSlot;     // multi-rank: per point slot into combA, or -1 (NULL on one rank)
    const int* posSlot;        // ... the same per smoothing-tile POSITION (k_smooth_halo: the regular tiles skip the shared points)
    // the shared points' OWN tiles (smgpu_halo_configure: SmoothTiles over the shared points only), per tile position: the point's
    // slot, the two-sharer point's peer code (the other rank's receive slot | this rank is the lower one << 30) or -1, its
    // number of send slots and the first of them (most shared points have one: no walk through sendOff / sendSlots)
    const int* spSlot; const int* spPeer; const int* spDst0; const int* spNDst;
    int* ownF;                 // != NULL (k_smooth_halo with its fix role): the local freeze flag of every shared point, by slot
    const double* combA;       // multi-rank: combined exchange-A records (13 doubles per shared point)
    // multi-rank, tiled kernels: a point with TWO sharers (nearly all shared points) is combined by the smoothing kernel
    // itself from the own and the received record (inlineCombine), and its freeze flag goes straight to its send slots
    // (inlinePackF): k_halo_combineA then only serves the few points with more sharers, k_halo_packF is not launched
    const int* combOff; const int* combSlots; const double* ownA; const double* recvA;
    const int* sendOff; const int* sendSlots; int* sendF;
    int inlineCombine, inlinePackF;
    // syncTools::syncPointList model of the minMagSqr / maxMagSqr folds (smgpu_set_sync_variant): 0 = the master's fold handed to
    // every sharer (globalMeshData::syncData), 1 = every sharer folds the others onto its own value
    int ownFold;
    // optional boundary layer treatment (layers.hpp): per point normal (re-normalised every iteration), hop count,
    // outer neighbour; per hop count the target edge length and the blending fraction
    double* layerNormal; const int* layerHops; const int* layerMap; const double* layerLen; const double* layerBlend;
    int lStride;               // multi-rank: doubles per exchange-L record in use (SMGPU_HALO_L_LAYERS or SMGPU_HALO_L_DOUBLES)
    const int* bndOfShared;    // multi-rank + boundary point smoothing: per shared point its index in the boundary tables or -1
    const double* combL;       // multi-rank + layers: per shared point the summed normals and the combined outer
                               // neighbour coordinates (6 doubles), or NULL
    PushView push;             // multi-rank, peer-store transport (smgpu_halo_set_push): all NULL = the host moves the records
    int* nActiveHost;          // pinned host word (or NULL): the end-of-iteration reduction leaves the iteration's nActive there,
                               // from which the host re-decides the walk's replay form (smgpu.hip:updateWalkMode)
    unsigned long long* nearTotal;   // [3] near-tie census since smgpu_create, by comparison: 0 = SM.C:923, 1 = SM.C:1367, 2 = SM.C:1391-1394 / 1421-1424
};

struct Prm {
    double maxStep, relStepFrac, minEdge;
    int totalMinFreeze;
    double smallAngle, largeAngle;   // M_PI * deg / 180.0  (SM.C:921, 1364-1365)
    float faCosLo, faCosHi;          // the f32 face-angle filter's thresholds on the cosine of an angle sum (kernels_filter.hpp)
    int layersOn;                    // boundary layer treatment enabled (SM.C:2024-2028)
    int bndOn;                       // boundary point smoothing enabled (SM.C:2080-2093): kernels_boundary.hpp
    long long nearUlps;              // near-tie window in ulp (default 4; SMGPU_NEARTIE_ULPS), see noteNear
};

// ---------------------------------------------------------------------------------------------
// SM.C:766-786 edgeEdgeAngle
__device__ __forceinline__ double clampAcos(double cosA) {
    const double MAXC = 0.99999;
    // std::max(-MAX, std::min(MAX, cosA)) with the std:: comparison forms (NaN -> +MAX)
    const double t = (cosA < MAXC) ? cosA : MAXC;
    const double c = (-MAXC < t) ? t : -MAXC;
    return smacos::acosX(c);      // (a fixed sequence of IEEE operations the checker can repeat bit for bit: smacos.hpp)
}
// Near-tie census.  The engine's acos (smacos.hpp) and glibc's -- the reference's -- differ in the last bit for ~6 % of the
// arguments, so a comparison of an angle with a threshold or with another angle (SM.C:923, 1367, 1391-1394, 1421-1424) whose two
// sides are only a few ulp apart could be decided the other way by the reference.  Every such comparison the exact kernels make
// is counted (sides with EQUAL bits are the same function of the same inputs on both sides and are not): per iteration in
// smgpu_iter_stats::nNearTies, since create in smgpu_get_near_ties.  Zero -- the normal case -- means no decision of the run hung
// on a last bit.  Elements the f32 filters decide are not counted: their margins are many orders of magnitude wider.
// (angles and thresholds are positive finite doubles: they order like their bit patterns)
__device__ __forceinline__ int nearTie(double a, double b, long long ulps) {
    const long long d = __double_as_longlong(a) - __double_as_longlong(b);
    return (d != 0 && (d < 0 ? -d : d) <= ulps) ? 1 : 0;
}
__device__ __forceinline__ void noteNear(const State& s, int cls, int n) {
    if (n) { atomicAdd(&s.acc->nNearTies, n); if (s.nearTotal) atomicAdd(&s.nearTotal[cls], (unsigned long long)n); }
}
// the comparisons of `(mn < small && mn < curMin) || (mx > large && mx > curMax)` the reference evaluates (short circuits included)
__device__ __forceinline__ int nearVerdict(double mn, double mx, double curMin, double curMax, const Prm& prm) {
    int n = nearTie(mn, prm.smallAngle, prm.nearUlps);
    bool first = false;
    if (mn < prm.smallAngle) { n += nearTie(mn, curMin, prm.nearUlps); first = mn < curMin; }
    if (!first) { n += nearTie(mx, prm.largeAngle, prm.nearUlps); if (mx > prm.largeAngle) n += nearTie(mx, curMax, prm.nearUlps); }
    return n;
}
__device__ __forceinline__ V3 unitTo(const V3& from, const V3& to) {
    const V3 v = to - from;
    return v / mag(v);
}

// ---------------------------------------------------------------------------------------------
// OpenFOAM primitiveMesh::makeFaceCentresAndAreas (.com v2412) -- one thread per face.
// Also emits the plain vertex average (calcFaceCenter SM.C:1103-1130 on current coordinates),
// which the face-angle pass reuses.
// foamOrg: OpenFOAM.org 12's formulas instead of OpenFOAM.com's (see geomFace in kernels_tiled.hpp)
__global__ void __launch_bounds__(kBlock) k_face_geom(MeshView m, State s, int wantAvg, int foamOrg) {
    if (s.acc->stop) return;
    const int f = blockIdx.x * kBlock + threadIdx.x;
    if (f >= m.nFaces) return;
    const int b = m.faceOff[f];
    const int n = m.faceOff[f + 1] - b;
    const double* __restrict__ P = s.ptsCur;
    V3 ctr, area;
    V3 fCentre = ldv(P, m.facePts[b]);
    for (int i = 1; i < n; ++i) fCentre = fCentre + ldv(P, m.facePts[b + i]);
    fCentre = fCentre / double(n);
    if (n == 3) {
        const V3 p0 = ldv(P, m.facePts[b]), p1 = ldv(P, m.facePts[b + 1]), p2 = ldv(P, m.facePts[b + 2]);
        ctr = (1.0 / 3.0) * ((p0 + p1) + p2);
        area = 0.5 * cross(p1 - p0, p2 - p0);
    } else if (foamOrg) {
        V3 sumA = v3(0, 0, 0);
        const V3 first = ldv(P, m.facePts[b]);
        V3 thisPoint = first;
        for (int i = 0; i < n; ++i) {
            const V3 nextPoint = (i == n - 1) ? first : ldv(P, m.facePts[b + i + 1]);
            sumA = sumA + cross(nextPoint - thisPoint, fCentre - thisPoint);
            thisPoint = nextPoint;
        }
        const V3 sumAHat = sumA / mag(sumA);
        double sumAn = 0.0;
        V3 sumAnc = v3(0, 0, 0);
        thisPoint = first;
        for (int i = 0; i < n; ++i) {
            const V3 nextPoint = (i == n - 1) ? first : ldv(P, m.facePts[b + i + 1]);
            const V3 a = cross(nextPoint - thisPoint, fCentre - thisPoint);
            const V3 c = (thisPoint + nextPoint) + fCentre;
            const double an = dot(a, sumAHat);
            sumAn += an;
            sumAnc = sumAnc + an * c;
            thisPoint = nextPoint;
        }
        ctr = (sumAn > SMGPU_VSMALL) ? ((1.0 / 3.0) * sumAnc) / sumAn : fCentre;
        area = 0.5 * sumA;
    } else {
        V3 sumN = v3(0, 0, 0), sumAc = v3(0, 0, 0);
        double sumA = 0.0;
        V3 thisPoint = ldv(P, m.facePts[b]);
        const V3 first = thisPoint;
        for (int i = 0; i < n; ++i) {
            const V3 nextPoint = (i == n - 1) ? first : ldv(P, m.facePts[b + i + 1]);
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
    stv(s.fCtr, f, ctr);
    stv(s.fArea, f, area);
    if (wantAvg) stv(s.fAvg, f, fCentre);
}

// OpenFOAM primitiveMesh::makeCellCentresAndVols (.com v2412) -- one thread per cell, faces in
// the accumulation order of the two forAll loops (owned ascending, then neighboured ascending).
__global__ void __launch_bounds__(kBlock) k_cell_centres(MeshView m, State s, int foamOrg) {
    if (s.acc->stop) return;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= m.nCells) return;
    const int b = m.cfOff[c], e = m.cfOff[c + 1];
    V3 cEst = v3(0, 0, 0);
    for (int k = b; k < e; ++k) cEst = cEst + ldv(s.fCtr, m.cfVal[k] & 0x7fffffff);
    cEst = cEst / double(e - b);
    V3 ctr = v3(0, 0, 0);
    double vol = 0.0;
    for (int k = b; k < e; ++k) {
        const int v = m.cfVal[k];
        const int f = v & 0x7fffffff;
        const V3 fc = ldv(s.fCtr, f);
        const V3 fA = ldv(s.fArea, f);
        double pyr3Vol = (v < 0) ? dot(fA, cEst - fc) : dot(fA, fc - cEst);
        if (foamOrg) pyr3Vol = (pyr3Vol > SMGPU_VSMALL) ? pyr3Vol : SMGPU_VSMALL;
        const V3 pc = (3.0 / 4.0) * fc + (1.0 / 4.0) * cEst;
        ctr = ctr + pyr3Vol * pc;
        vol += pyr3Vol;
    }
    if (fabs(vol) > SMGPU_VSMALL) ctr = ctr / vol;
    else ctr = cEst;
    stv(s.cellCtr, c, ctr);
}

// ---------------------------------------------------------------------------------------------
// Per-point record of the local part of centroidalSmoothing + findClosestPoints.
struct PointLocal {
    V3 sum; int count;       // SM.C:121-130
    V3 r1, r2, r3; int hc;   // SM.C:325-387
};

__device__ __forceinline__ bool shareCell(const MeshView& m, int q1, int q2) {
    // (q2 in pointNeighPoints[q1]) <=> pointCells(q1) and pointCells(q2) intersect (SM.C:196-214, 383)
    int a = m.pcOff[q1], ae = m.pcOff[q1 + 1];
    int b = m.pcOff[q2], be = m.pcOff[q2 + 1];
    if (a >= ae || b >= be) return false;
    int ca = m.pcVal[a], cb = m.pcVal[b];
    while (true) {
        if (ca == cb) return true;
        if (ca < cb) { if (++a >= ae) return false; ca = m.pcVal[a]; }
        else { if (++b >= be) return false; cb = m.pcVal[b]; }
    }
}

__device__ __forceinline__ void pointLocal(const MeshView& m, const State& s, int p, const V3& cur, bool internal,
                                           PointLocal& L, int& err, bool centroidAll = false) {
    // SM.C:116-130: only internal points gather unless doBoundarySmoothing (centroidAll)
    L.sum = v3(0, 0, 0);
    L.count = 0;
    if (internal || centroidAll) {
        const int b = m.pcOff[p], e = m.pcOff[p + 1];
        L.count = e - b;
        for (int k = b; k < e; ++k) L.sum = L.sum + ldv(s.cellCtr, m.pcVal[k]);
    }
    // SM.C:325-352: three shortest incident edges in stable (length, list position) order;
    // boundary points only look at boundary neighbours (SM.C:294)
    double l1 = 0, l2 = 0, l3 = 0;
    int q1 = -1, q2 = -1, q3 = -1;
    const int b = m.ppOff[p], e = m.ppOff[p + 1];
    for (int k = b; k < e; ++k) {
        const int q = m.ppPt[k];
        if (!internal && (m.pflags[q] & PF_INTERNAL)) continue;
        const double len = mag(cur - ldv(s.ptsCur, q));  // getPointDistance(neigh, cCoords)
        if (q1 < 0 || len < l1) { l3 = l2; q3 = q2; l2 = l1; q2 = q1; l1 = len; q1 = q; }
        else if (q2 < 0 || len < l2) { l3 = l2; q3 = q2; l2 = len; q2 = q; }
        else if (q3 < 0 || len < l3) { l3 = len; q3 = q; }
    }
    if (q2 < 0) { err = 1; L.r1 = L.r2 = L.r3 = v3(0, 0, 0); L.hc = 0; return; }  // SM.C:354-362
    L.r1 = ldv(s.ptsCur, q1) - cur;
    L.r2 = ldv(s.ptsCur, q2) - cur;
    L.r3 = (q3 < 0) ? v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT) : ldv(s.ptsCur, q3) - cur;  // SM.C:373-380
    L.hc = shareCell(m, q1, q2) ? 1 : 0;
}

// SM.C:489-543.  m1..m3 = mag(closestPoint1..3) (SM.C:509-510), passed in because the callers already
// hold them: |x_q - x_p| computed as an edge length (SM.C:336) is the same value bit for bit.
__device__ __forceinline__ double arRatioLen(const V3& c1, const V3& c2, double m1, double m2, double m3, bool hasCommonCell,
                                             bool internal) {
    if (hasCommonCell) return 0.0;
    const V3 z = v3(0, 0, 0);
    if ((c1 == z) || (c2 == z)) return 0.0;
    const double lengthRatio1 = m2 / m1;
    const double lengthRatio2 = m3 / m2;
    if (internal) {
        const double minRatio = 1.5, maxRatio = 3.0;
        if ((lengthRatio1 < minRatio) && (lengthRatio2 > minRatio)) {
            const double frac = (lengthRatio2 - minRatio) / (maxRatio - minRatio);
            const double t = (0.0 > frac) ? 0.0 : frac;   // Foam::max
            return (1.0 < t) ? 1.0 : t;                   // Foam::min
        }
        return 0.0;
    } else {
        const double minRatio = 1.0, maxRatio = 2.0;
        const double frac = (lengthRatio1 - minRatio) / (maxRatio - minRatio);
        const double t = (0.0 > frac) ? 0.0 : frac;
        return (1.0 < t) ? 1.0 : t;
    }
}
__device__ __forceinline__ double arRatio(const V3& c1, const V3& c2, const V3& c3, bool hasCommonCell, bool internal) {
    return arRatioLen(c1, c2, mag(c1), mag(c2), mag(c3), hasCommonCell, internal);
}

// v / double(n) for a small positive count: a power-of-two count is an exact scaling, so the product
// with the exact reciprocal is the same correctly rounded value as the quotient, at a tenth of the cost
__device__ __forceinline__ V3 divByCount(const V3& v, int n) {
    if ((n & (n - 1)) == 0) { const double r = 1.0 / double(n); return v3(v.x * r, v.y * r, v.z * r); }
    return v / double(n);
}

// residual = max over points (SM.C:1556-1565), nFrozenPoints = count (SM.C:2384-2392): every workgroup
// publishes one partial (no same-address atomics: thousands of workgroups on one word serialise at
// ~90 atomics/us); k_finish reduces the partials.
template <int T>
__device__ __forceinline__ void blockPublish(const State& s, double dist, int frozenCount, int partialSlot) {
    __shared__ double shMax[T / 64];
    __shared__ int shCnt[T / 64];
    if (!(dist > 0.0)) dist = 0.0;  // NaN never wins "distance > maxStep"
    for (int o = 32; o > 0; o >>= 1) {
        const double od = __shfl_down(dist, o, 64);
        const int oc = __shfl_down(frozenCount, o, 64);
        dist = (od > dist) ? od : dist;
        frozenCount += oc;
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { shMax[w] = dist; shCnt[w] = frozenCount; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double d = shMax[0]; int c = shCnt[0];
        for (int i = 1; i < T / 64; ++i) { d = (shMax[i] > d) ? shMax[i] : d; c += shCnt[i]; }
        s.blkMax[partialSlot] = d;
        s.blkCnt[partialSlot] = c;
    }
}

// Optional boundary layer treatment of one point, between the first step clamp and restrictEdgeShortening
// (SM.C:2283-2305), serial run:
//  * SM.C:2266 calculateBoundaryPointNormals -- of its effects only the last pass matters here (OBB.C:222-229): every
//    non-zero normal is divided by its magnitude again, each iteration (internal points carry copies made at set-up;
//    the normals of boundary points are not read by the treatment);
//  * updateNeighCoords OBB.C:464-500: the outer neighbour's CURRENT coordinates;
//  * blendWithOrthogonalPoints OBB.C:507-567, with the hop-count functions tabulated by the host (layers.cpp);
//  * constrainMaxStepLength once more, for every point (SM.C:2304).
// Under -parallel a shared point takes the plusEq-synchronised normal (the sum over its sharers, OBB.C:184-190) and the
// minMagSqr-synchronised neighbour coordinates (OBB.C:490-496) from combL instead of its local values.
__device__ __forceinline__ V3 layerTreat(const State& s, const Prm& prm, int p, bool internal, const V3& cur, V3 np) {
    const int slot = (s.combL && s.sharedSlot) ? s.sharedSlot[p] : -1;
    const double* cl = (slot >= 0) ? s.combL + (size_t)slot * s.lStride : nullptr;
    V3 n = cl ? v3(cl[0], cl[1], cl[2]) : ldv(s.layerNormal, p);
    const V3 z = v3(0, 0, 0);
    if (n != z) {
        // with boundary point smoothing k_bnd_normals has already recomputed the boundary points' normals this iteration
        if (!(prm.bndOn && !internal)) {
            n = n / mag(n);
            stv(s.layerNormal, p, n);
        }
        const int hops = s.layerHops[p];
        if (internal && hops >= 1) {
            const V3 outer = cl ? v3(cl[3], cl[4], cl[5]) : ldv(s.ptsCur, s.layerMap[p]);
            const double blendFrac = s.layerBlend[hops];
            const V3 orthoPoint = outer + s.layerLen[hops] * n;
            np = blendFrac * orthoPoint + (1.0 - blendFrac) * np;
        }
    }
    const V3 stepDir = np - cur;
    const double len = mag(stepDir);
    const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
    return cur + (prm.relStepFrac * globalScale) * stepDir;
}

// The fused per-point proposal kernel: centroidalSmoothing SM.C:96-166, aspectRatioSmoothing
// SM.C:548-593 (+ findClosestPoints/calcARSmoothingRatio), constrainMaxStepLength SM.C:684-754,
// restrictEdgeShortening SM.C:602-652.  FINAL = true additionally does restore/count SM.C:2384-2392,
// residual SM.C:1546-1565 and movePoints (write into the other coordinate buffer); valid only when
// no later evaluator reads neighbours' proposals (angle constraints off, one rank).
template <bool FINAL>
__global__ void __launch_bounds__(kBlock) k_smooth(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    double dist = 0.0;
    int fcount = 0;
    if (p < m.nPoints) {
        const uint8_t fl = m.pflags[p];
        const bool internal = fl & PF_INTERNAL;
        const V3 cur = ldv(s.ptsCur, p);
        PointLocal L;
        int err = 0;
        const int slot = s.sharedSlot ? s.sharedSlot[p] : -1;
        if (slot >= 0) {
            // multi-rank: values already combined over all sharing ranks (syncPointList, SM.C:134-148,391-478)
            const double* r = s.combA + (size_t)slot * SMGPU_HALO_A_DOUBLES;
            L.sum = v3(r[0], r[1], r[2]);
            L.r1 = v3(r[3], r[4], r[5]);
            L.r2 = v3(r[6], r[7], r[8]);
            L.r3 = v3(r[9], r[10], r[11]);
            const long long pk = __double_as_longlong(r[12]);
            L.count = (int)(pk & 0xffffffffll);
            L.hc = (int)(pk >> 32);
        } else {
            pointLocal(m, s, p, cur, internal, L, err, prm.bndOn != 0);
            if (err) s.acc->err = 1;
        }
        // SM.C:155-163
        V3 np = cur;
        if (L.count) np = L.sum / double(L.count);
        // SM.C:580-590
        const double blendFrac = arRatio(L.r1, L.r2, L.r3, L.hc != 0, internal);
        if (blendFrac > 0.0) {
            const V3 aCoords = cur + (L.r1 + L.r2) / 2.0;
            np = (1.0 - blendFrac) * np + blendFrac * aCoords;
        }
        // SM.C:722-745
        {
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        if (prm.layersOn) np = layerTreat(s, prm, p, internal, cur, np);   // SM.C:2283-2305
        const bool deferBnd = prm.bndOn && !internal;   // k_bnd_fix finishes the boundary points (kernels_boundary.hpp)
        if (prm.bndOn && internal) {                                       // SM.C:2356
            const V3 stepDir = np - cur;
            const double len = mag(stepDir);
            const double globalScale = (len > prm.maxStep) ? prm.maxStep / (len * prm.relStepFrac) : 1.0;
            np = cur + (prm.relStepFrac * globalScale) * stepDir;
        }
        // SM.C:611-648 (isFrozenPoint is all false here: reset at SM.C:2262)
        bool frozen = false;
        {
            double shortestCur = SMGPU_GREAT, shortestNew = SMGPU_GREAT;
            const int b = m.ppOff[p], e = m.ppOff[p + 1];
            for (int k = b; k < e; ++k) {
                const V3 nb = ldv(s.ptsCur, m.ppPt[k]);
                const double tc = mag(cur - nb);
                if (tc < shortestCur) shortestCur = tc;
                const double tn = mag(np - nb);
                if (tn < shortestNew) shortestNew = tn;
            }
            const double shortest = (shortestNew < shortestCur) ? shortestNew : shortestCur;
            if (prm.totalMinFreeze && (shortest < prm.minEdge)) frozen = true;
            else if ((shortestNew < prm.minEdge) && (shortestNew < shortestCur)) frozen = true;
        }
        if (deferBnd) stv(s.prop, p, np);
        else if (FINAL) {
            if (frozen || (!internal && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
            dist = mag(np - cur) / prm.maxStep;
            stv(s.ptsNext, p, np);
        } else {
            stv(s.prop, p, np);
            s.frozen[p] = frozen ? 1 : 0;
        }
    }
    if (FINAL) blockPublish<kBlock>(s, dist, fcount, blockIdx.x);
}

// restrictMinEdgeAngleDecrease SM.C:900-930 (+ calc_min_edge_angles :837-894) -- one thread per point.
// The six distinct unit vectors of the five edgeEdgeAngle calls are each formed once (same
// operations, same results as forming them per call).
__global__ void __launch_bounds__(kBlock) k_edge_angle(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= m.nPoints) return;
    if (s.frozen[p]) return;
    const V3 cp0 = ldv(s.ptsCur, p);
    const V3 np0 = ldv(s.prop, p);
    double minC = 1.7976931348623157e308, minN = 1.7976931348623157e308;  // DBL_MAX
    const int b = m.pfOff[p], e = m.pfOff[p + 1];
    for (int k = b; k < e; ++k) {
        const int a1 = m.pfPrev[k], a2 = m.pfNext[k];
        const V3 cp1 = ldv(s.ptsCur, a1), cp2 = ldv(s.ptsCur, a2);
        const V3 np1 = ldv(s.prop, a1), np2 = ldv(s.prop, a2);
        const double cAngle = clampAcos(dot(unitTo(cp0, cp1), unitTo(cp0, cp2)));
        const V3 uc1 = unitTo(np0, cp1), uc2 = unitTo(np0, cp2), un1 = unitTo(np0, np1), un2 = unitTo(np0, np2);
        const double nAngle0 = clampAcos(dot(uc1, uc2));
        const double nAngle1 = clampAcos(dot(un1, un2));
        const double nAngle2 = clampAcos(dot(uc1, un2));
        const double nAngle3 = clampAcos(dot(un1, uc2));
        double nAngle = (nAngle0 < nAngle1) ? nAngle0 : nAngle1;
        nAngle = (nAngle < nAngle2) ? nAngle : nAngle2;
        nAngle = (nAngle < nAngle3) ? nAngle : nAngle3;
        if (cAngle < minC) minC = cAngle;
        if (nAngle < minN) minN = nAngle;
    }
    if ((minN < prm.smallAngle) && (minN < minC)) s.frozen[p] = 1;
    noteNear(s, 0, nearTie(minN, prm.smallAngle, prm.nearUlps) + ((minN < prm.smallAngle) ? nearTie(minN, minC, prm.nearUlps) : 0));
}

// restrictMinEdgeAngleDecrease SM.C:900-930, wave-cooperative form: 4 lanes per point (measured best of 2, 4, 8).
//  * every incident edge (p, q) appears in several of p's faces (4 in a hex mesh), so the three unit
//    vectors per neighbour q -- (x_q - x_p)/|.|, (x_q - n_p)/|.|, (n_q - n_p)/|.| -- are formed ONCE per
//    neighbour (lane j takes neighbour j) and shared through LDS instead of once per (face, corner);
//  * the lanes then take the point's faces (5 dot products each) and keep the LARGEST clamped cosine;
//    acos is evaluated only on the two extremes: min_i acos(c_i) = acos(max_i c_i) because acos is
//    monotone (SM.C:880-886 take minima of angles; same operations on the same operands otherwise).
constexpr int kEaLanes = 4;
constexpr int kEaPointsPerBlock = kBlock / kEaLanes;

__device__ __forceinline__ double clampCos(double cosA) {
    const double MAXC = 0.99999;   // SM.C:781-782, std::min/std::max comparison forms (NaN -> +MAX)
    const double t = (cosA < MAXC) ? cosA : MAXC;
    return (-MAXC < t) ? t : -MAXC;
}

__global__ void __launch_bounds__(kBlock) k_edge_angle_coop(MeshView m, State s, Prm prm, int maxEntries, const uint8_t* eaMaybe) {
    if (s.acc->stop) return;
    if (eaMaybe && s.acc->nEaMaybe == 0) return;   // the filter decided every point
    extern __shared__ double lds[];
    double* U = lds;   // 9 arrays of maxEntries: ucc.xyz, uc.xyz, un.xyz
    const int tid = threadIdx.x, g = tid & (kEaLanes - 1);
    const int pBase = blockIdx.x * kEaPointsPerBlock;
    const int p = pBase + (tid / kEaLanes);
    const bool valid = p < m.nPoints;
    // eaMaybe (may be NULL): points the f32 filter (kernels_filter.hpp) could not rule out
    const bool active = valid && !s.frozen[p] && (!eaMaybe || eaMaybe[p]);
    if (!__syncthreads_or(active ? 1 : 0)) return;   // nothing to decide in this workgroup
    const int e0 = m.ppOff[pBase];
    int nb = 0, nv = 0;
    V3 cp0 = v3(0, 0, 0), np0 = v3(0, 0, 0);
    if (active) {
        nb = m.ppOff[p]; nv = m.ppOff[p + 1] - nb;
        cp0 = ldv(s.ptsCur, p); np0 = ldv(s.prop, p);
        for (int j = g; j < nv; j += kEaLanes) {
            const int q = m.ppPt[nb + j];
            const V3 cq = ldv(s.ptsCur, q), nq = ldv(s.prop, q);
            const V3 a = unitTo(cp0, cq), b = unitTo(np0, cq), c = unitTo(np0, nq);
            const int i = nb + j - e0;
            U[i] = a.x; U[maxEntries + i] = a.y; U[2 * maxEntries + i] = a.z;
            U[3 * maxEntries + i] = b.x; U[4 * maxEntries + i] = b.y; U[5 * maxEntries + i] = b.z;
            U[6 * maxEntries + i] = c.x; U[7 * maxEntries + i] = c.y; U[8 * maxEntries + i] = c.z;
        }
    }
    __syncthreads();
    double maxC = -2.0, maxN = -2.0;
    int nf = 0;
    if (active) {
        const int fb = m.pfOff[p];
        nf = m.pfOff[p + 1] - fb;
        for (int k = g; k < nf; k += kEaLanes) {
            const int sa = m.pfPrevSlot[fb + k], sb = m.pfNextSlot[fb + k];
            V3 cca, ccb, ca, cb, na, nbv;
            if (sa != 255 && sb != 255) {
                const int ia = nb + sa - e0, ib = nb + sb - e0;
                cca = v3(U[ia], U[maxEntries + ia], U[2 * maxEntries + ia]);
                ca = v3(U[3 * maxEntries + ia], U[4 * maxEntries + ia], U[5 * maxEntries + ia]);
                na = v3(U[6 * maxEntries + ia], U[7 * maxEntries + ia], U[8 * maxEntries + ia]);
                ccb = v3(U[ib], U[maxEntries + ib], U[2 * maxEntries + ib]);
                cb = v3(U[3 * maxEntries + ib], U[4 * maxEntries + ib], U[5 * maxEntries + ib]);
                nbv = v3(U[6 * maxEntries + ib], U[7 * maxEntries + ib], U[8 * maxEntries + ib]);
            } else {   // neighbour not in the slot table (valence > 254): form the vectors directly
                const int a1 = m.pfPrev[fb + k], a2 = m.pfNext[fb + k];
                const V3 cp1 = ldv(s.ptsCur, a1), cp2 = ldv(s.ptsCur, a2), np1 = ldv(s.prop, a1), np2 = ldv(s.prop, a2);
                cca = unitTo(cp0, cp1); ccb = unitTo(cp0, cp2);
                ca = unitTo(np0, cp1); cb = unitTo(np0, cp2);
                na = unitTo(np0, np1); nbv = unitTo(np0, np2);
            }
            const double cC = clampCos(dot(cca, ccb));
            const double c0 = clampCos(dot(ca, cb));      // nAngle0: (n0; x1, x2)
            const double c1 = clampCos(dot(na, nbv));     // nAngle1: (n0; n1, n2)
            const double c2 = clampCos(dot(ca, nbv));     // nAngle2: (n0; x1, n2)
            const double c3 = clampCos(dot(na, cb));      // nAngle3: (n0; n1, x2)
            double cN = (c0 > c1) ? c0 : c1;
            cN = (cN > c2) ? cN : c2;
            cN = (cN > c3) ? cN : c3;
            if (cC > maxC) maxC = cC;
            if (cN > maxN) maxN = cN;
        }
    }
    for (int o = kEaLanes / 2; o > 0; o >>= 1) {
        const double oc = __shfl_xor(maxC, o, 64), on = __shfl_xor(maxN, o, 64);
        maxC = (oc > maxC) ? oc : maxC;
        maxN = (on > maxN) ? on : maxN;
    }
    // one acos evaluation per wave serves both extremes: lane 0 of each group takes maxC, lane 1 maxN
    const double ang = smacos::acosX((g == 0) ? maxC : maxN);
    const double minN = __shfl_down(ang, 1, 64);   // lane 0 of the group reads lane 1's value
    if (active && g == 0 && nf > 0) {
        const double minC = ang;
        if ((minN < prm.smallAngle) && (minN < minC)) s.frozen[p] = 1;
        noteNear(s, 0, nearTie(minN, prm.smallAngle, prm.nearUlps) + ((minN < prm.smallAngle) ? nearTie(minN, minC, prm.nearUlps) : 0));
    }
}

// ---------------------------------------------------------------------------------------------
// calcMinMaxFaceAngleForEdge SM.C:1135-1231 for one edge, with optional substitution of the
// coordinates of two points (i1 -> c1, i2 -> c2; i = -1 disables).  SUBST = false is the
// current-mesh form (SM.C:1235-1247) and uses the per-face vertex averages from k_face_geom.
// vertex average of a face of the current coordinates (calcFaceCenter SM.C:1103-1130; the same operations as the
// geometry kernels' fCentre, so the same bits)
__device__ __forceinline__ V3 faceAverage(const MeshView& m, const State& s, int f) {
    const int b = m.faceOff[f], n = m.faceOff[f + 1] - b;
    V3 fc = ldv(s.ptsCur, m.facePts[b]);
    for (int i = 1; i < n; ++i) fc = fc + ldv(s.ptsCur, m.facePts[b + i]);
    return divByCount(fc, n);
}

template <bool SUBST>
__device__ __forceinline__ void edgeFaceAngles(const MeshView& m, const State& s, int e, int i1, const V3& c1, int i2,
                                               const V3& c2, double& minA, double& maxA) {
    const int e0I = m.edges[2 * e], e1I = m.edges[2 * e + 1];
    V3 e0 = ldv(s.ptsCur, e0I), e1 = ldv(s.ptsCur, e1I);
    if (SUBST) {
        if (i1 >= 0 && e0I == i1) e0 = c1; else if (i2 >= 0 && e0I == i2) e0 = c2;
        if (i1 >= 0 && e1I == i1) e1 = c1; else if (i2 >= 0 && e1I == i2) e1 = c2;
    }
    const V3 cC = 0.5 * (e0 + e1);
    const V3 d = e1 - e0;
    const V3 eVec = d / mag(d);
    const int fb = m.efOff[e];
    auto faceVec = [&](int f) -> V3 {
        V3 fc;
        if (SUBST) {
            // calcFaceCenter SM.C:1103-1130
            fc = v3(0, 0, 0);
            const int b = m.faceOff[f], n = m.faceOff[f + 1] - b;
            for (int i = 0; i < n; ++i) {
                const int q = m.facePts[b + i];
                if (i1 >= 0 && q == i1) fc = fc + c1;
                else if (i2 >= 0 && q == i2) fc = fc + c2;
                else fc = fc + ldv(s.ptsCur, q);
            }
            fc = fc / double(n);
        } else {
            fc = faceAverage(m, s, f);
        }
        const V3 cf = cC - fc;
        const double dp = dot(cf, eVec);
        const V3 pC = fc + dp * eVec;
        const V3 w = pC - cC;
        return w / mag(w);
    };
    minA = 2.0 * SMGPU_PI;
    maxA = 0.0;
    const int cb = m.ecOff[e], ce = m.ecOff[e + 1];
    for (int k = cb; k < ce; ++k) {
        const V3 p0 = faceVec(m.efFace[fb + m.ecF0[k]]);
        const V3 p1 = faceVec(m.efFace[fb + m.ecF1[k]]);
        const V3 cc = ldv(s.cellCtr, m.ecCell[k]);  // mesh.C()[cellI] of the CURRENT mesh, SM.C:1218
        const V3 cf = cC - cc;
        const double dp = dot(cf, eVec);
        const V3 pC = cc + dp * eVec;
        const V3 w = pC - cC;
        const V3 cV = w / mag(w);
        // calcEdgeCenterEdgeAngle SM.C:980-998
        const double angle = clampAcos(dot(p0, cV)) + clampAcos(dot(cV, p1));
        if (angle < minA) minA = angle;
        if (angle > maxA) maxA = angle;
    }
}

// calcCurrentMinMaxFaceAnglesForEdges SM.C:1252-1270 -- one thread per edge.  Faces and cells are
// visited in ring order around the edge (Topology::ringFace): cell i sits between ring faces i and i+1,
// so each projected face-centre vector is formed once and handed on (the reference forms them per
// edge face too, SM.C:1183-1200; min/max over the cells do not depend on the visiting order).
// faMaybe (may be NULL): only edges with an end point the filter could not rule out are evaluated.
__device__ __forceinline__ void faEdgeExact(const MeshView& m, const State& s, int e) {
    double mn, mx;
    if (!m.edgeRingOk[e]) {
        const V3 z = v3(0, 0, 0);
        edgeFaceAngles<false>(m, s, e, -1, z, -1, z, mn, mx);
    } else {
        const V3 e0 = ldv(s.ptsCur, m.edges[2 * e]), e1 = ldv(s.ptsCur, m.edges[2 * e + 1]);
        const V3 cC = 0.5 * (e0 + e1);
        const V3 d = e1 - e0;
        const V3 eVec = d / mag(d);
        auto project = [&](const V3& c) -> V3 {   // SM.C:1189-1196 / 1219-1223
            const V3 cf = cC - c;
            const double dp = dot(cf, eVec);
            const V3 pC = c + dp * eVec;
            const V3 w = pC - cC;
            return w / mag(w);
        };
        const int fb = m.efOff[e], nf = m.efOff[e + 1] - fb;
        const int cb = m.ecOff[e], nc = m.ecOff[e + 1] - cb;
        const V3 first = project(faceAverage(m, s, m.ringFace[fb]));
        V3 prev = first;
        mn = 2.0 * SMGPU_PI;
        mx = 0.0;
        for (int i = 0; i < nc; ++i) {
            const V3 next = (i + 1 < nf) ? project(faceAverage(m, s, m.ringFace[fb + i + 1])) : first;
            const V3 cV = project(ldv(s.cellCtr, m.ringCell[cb + i]));   // mesh.C()[cellI], SM.C:1218
            const double angle = clampAcos(dot(prev, cV)) + clampAcos(dot(cV, next));   // SM.C:980-998
            if (angle < mn) mn = angle;
            if (angle > mx) mx = angle;
            prev = next;
        }
    }
    s.edgeMin[e] = mn;
    s.edgeMax[e] = mx;
}
__global__ void __launch_bounds__(kBlock) k_fa_edges(MeshView m, State s, const uint8_t* faMaybe) {
    if (s.acc->stop) return;
    if (faMaybe && s.acc->nFaMaybe == 0) return;   // the filter found every edge inside the good range
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= m.nEdges) return;
    if (faMaybe && faMaybe[m.edges[2 * e]] != s.faGen && faMaybe[m.edges[2 * e + 1]] != s.faGen) return;
    faEdgeExact(m, s, e);
}

// ---- ordered compaction of marked points without a scan launch ------------------------------------------------------------
// A workgroup owns a chunk of kChunk = 4096 consecutive points, 16 per thread: the thread's mark bytes are ONE 16-byte load
// (round 2 used one thread per point: 10 M threads each loading a byte, shuffling and synchronising -- 146 + 255 us for
// k_fa_list_count / fill on the 10 M-point mesh, plus a single-workgroup scan launch over 40 k block counts, 76 us).  A count
// launch leaves two numbers per chunk (marked points, entries they list); the fill launch computes its own offsets: every
// workgroup sums the counts of the chunks before it (2 479 of them on a 10 M-point mesh: ten loads per thread) -- what a scan
// launch in between would deliver, without the launch and without any cross-workgroup hand-off inside a kernel.  The last
// workgroup leaves the totals.  Lists come out in ascending point order.
constexpr int kChunkPer = 16;
constexpr int kChunk = kBlock * kChunkPer;
inline int chunkGrid(int64_t n) { return (int)((n + kChunk - 1) / kChunk); }
// the 16 mark bytes of this thread's points (zeros beyond the end; the mark arrays are allocated with 16 spare bytes)
__device__ __forceinline__ uint4 chunkMarks(const uint8_t* marks, int base, int n) {
    return (base < n) ? *reinterpret_cast<const uint4*>(marks + base) : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ unsigned chunkByte(const uint4& q, int i) {
    const unsigned wsel = (i >> 2) == 0 ? q.x : (i >> 2) == 1 ? q.y : (i >> 2) == 2 ? q.z : q.w;
    return (wsel >> (8 * (i & 3))) & 0xffu;
}
// sums over the workgroup; every thread gets them
__device__ __forceinline__ void chunkReduce2(int& a, int& e) {
    __shared__ int sh[2 * (kBlock / 64)];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); e += __shfl_xor(e, o, 64); }
    __syncthreads();   // (a previous use of sh has been read)
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[kBlock / 64 + (threadIdx.x >> 6)] = e; }
    __syncthreads();
    a = e = 0;
    for (int i = 0; i < kBlock / 64; ++i) { a += sh[i]; e += sh[kBlock / 64 + i]; }
}
// sums of the chunk counts of the workgroups [0, b) and -- all != 0 -- of all nBlk workgroups
__device__ __forceinline__ void chunkPrefix2(const int* blkA, const int* blkE, int b, int nBlk, bool all, int& pa, int& pe, int& ta, int& te) {
    int a = 0, e = 0, a2 = 0, e2 = 0;
    for (int i = threadIdx.x; i < (all ? nBlk : b); i += kBlock) {
        const int va = blkA[i], ve = blkE[i];
        if (i < b) { a += va; e += ve; }
        a2 += va; e2 += ve;
    }
    chunkReduce2(a, e);
    pa = a; pe = e;
    if (all) { chunkReduce2(a2, e2); ta = a2; te = e2; }
}
// exclusive scan of (a, e) over the workgroup's threads in thread order
__device__ __forceinline__ void chunkScan2(int a, int e, int& xa, int& xe) {
    __shared__ int sw[2 * (kBlock / 64)];
    int ia = a, ie = e;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(ia, o, 64), te = __shfl_up(ie, o, 64);
        if (lane >= o) { ia += ta; ie += te; }
    }
    __syncthreads();
    if (lane == 63) { sw[wv] = ia; sw[kBlock / 64 + wv] = ie; }
    __syncthreads();
    xa = ia - a; xe = ie - e;
    for (int k = 0; k < wv; ++k) { xa += sw[k]; xe += sw[kBlock / 64 + k]; }
}

// exclusive scan of one value over the workgroup's threads in thread order, and the workgroup's total
__device__ __forceinline__ void chunkScan1(int v, int& x, int& total) {
    __shared__ int sw1[kBlock / 64];
    int iv = v;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(iv, o, 64);
        if (lane >= o) iv += t;
    }
    __syncthreads();
    if (lane == 63) sw1[wv] = iv;
    __syncthreads();
    x = iv - v;
    total = 0;
    for (int k = 0; k < kBlock / 64; ++k) { if (k < wv) x += sw1[k]; total += sw1[k]; }
}
// The chunk's marked points in ascending order as chunk-local offsets in LDS: list[0 .. n); returns n to every thread.  The
// work per marked point (gathers over its neighbours) is then dealt one point per thread -- marked points come in clusters
// (a refinement interface), and a thread walking its own 16 points one after the other was a chain of up to 16 dependent
// gather loops while the other lanes of its wave waited.
__device__ __forceinline__ int chunkCompact(const uint4& q, unsigned gen, int base, int nPoints, uint16_t* list) {
    int a = 0;
    if (q.x | q.y | q.z | q.w)
        for (int i = 0; i < kChunkPer; ++i) a += (chunkByte(q, i) == gen && base + i < nPoints) ? 1 : 0;
    int x, n;
    chunkScan1(a, x, n);
    if (a)
        for (int i = 0; i < kChunkPer; ++i)
            if (chunkByte(q, i) == gen && base + i < nPoints) list[x++] = (uint16_t)(threadIdx.x * kChunkPer + i);
    __syncthreads();
    return n;
}

// With the filter on, what needs the exact evaluation is sparse (the end points of the UNSURE edges and all their edges:
// 2 % of a 10 M-cell refinement-interface mesh, nothing on a good hex block), but one thread per edge asking "is one of my
// end points marked?" costs two random byte gathers for each of the 30 M edges (0.56 ms).  So the marked points list
// themselves and their edges (every edge once: by its lower marked end point), and the exact kernels run on the lists.
// Listing without atomics (a returning atomic on one counter word runs at ~90 per microsecond chip-wide, wave-aggregated or
// not: 2 ms for this mesh): count per chunk (k_fa_list_count), then every chunk writes its ids at its offsets
// (k_fa_list_fill, which leaves the totals in acc->nFaPts / nFaEdges).  An edge is listed by its marked end point -- the lower
// one when both are; the lists come out in point order, so the exact kernel's gathers stay local.
__device__ __forceinline__ int faListedEdges(const MeshView& m, const State& s, const uint8_t* faMaybe, int p, int* out) {
    int n = 0;
    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
        const int q = m.ppPt[k];
        if ((faMaybe[q] != s.faGen) || (p < q)) { if (out) out[n] = m.peEdge[k]; ++n; }
    }
    return n;
}
__global__ void __launch_bounds__(kBlock) k_fa_list_count(MeshView m, State s, const uint8_t* faMaybe, int* blkA, int* blkE) {
    if (s.acc->stop) return;
    if (s.acc->nFaMaybe == 0) return;                  // the filter found every edge inside the good range: no lists
    __shared__ uint16_t list[kChunk];
    const int cb = blockIdx.x * kChunk, base = cb + threadIdx.x * kChunkPer;
    const int n = chunkCompact(chunkMarks(faMaybe, base, m.nPoints), s.faGen, base, m.nPoints, list);
    int a = 0, e = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) e += faListedEdges(m, s, faMaybe, cb + list[i], nullptr);
    if (n > 0) chunkReduce2(a, e);
    if (threadIdx.x == 0) { blkA[blockIdx.x] = n; blkE[blockIdx.x] = e; }
}
__global__ void __launch_bounds__(kBlock) k_fa_list_fill(MeshView m, State s, const uint8_t* faMaybe, const int* blkA, const int* blkE) {
    if (s.acc->stop) return;
    if (s.acc->nFaMaybe == 0) return;                  // nFaPts / nFaEdges stay 0 (reset at the end of every iteration)
    __shared__ uint16_t list[kChunk];
    const int cb = blockIdx.x * kChunk, base = cb + threadIdx.x * kChunkPer;
    const int n = chunkCompact(chunkMarks(faMaybe, base, m.nPoints), s.faGen, base, m.nPoints, list);
    int pa, pe, ta, te;
    chunkPrefix2(blkA, blkE, (int)blockIdx.x, (int)gridDim.x, false, pa, pe, ta, te);
    int running = 0;
    for (int r = 0; r < n; r += kBlock) {              // (n is the same for every thread)
        const int i = r + threadIdx.x;
        const int p = (i < n) ? cb + list[i] : -1;
        const int ne = (p >= 0) ? faListedEdges(m, s, faMaybe, p, nullptr) : 0;
        int xe, tot;
        chunkScan1(ne, xe, tot);
        if (p >= 0) {
            s.faPointList[pa + i] = p;
            (void)faListedEdges(m, s, faMaybe, p, s.faEdgeList + (pe + running + xe));
        }
        running += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { s.acc->nFaPts = pa + n; s.acc->nFaEdges = pe + running; }
}
__global__ void __launch_bounds__(kBlock) k_fa_edges_list(MeshView m, State s) {
    if (s.acc->stop) return;
    const int n = s.acc->nFaEdges;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) faEdgeExact(m, s, s.faEdgeList[i]);
}

// (Round 4, measured and removed: the same job with one lane per RING PLACE -- a wave takes up to 16 listed edges whose rings fit
// its 64 lanes, lane (edge, place) forms one face vector and one cell vector, segmented min / max.  Bit-equal, but no faster:
// timed ALONE the list kernels take 165 us for the ~600 k listed edges of the 10 M-cell cavity mesh either way (the 574 us of round
// 3's kernel trace were measured while the proposal kernel shared the chip), profiles/r4/.)
// mapCurrentMinMaxFaceAnglesToPoints SM.C:938-975 as a gather over pointEdges, plus the
// good-range test SM.C:1367-1369 -- one thread per point.
__device__ __forceinline__ void faPointMinMax(const MeshView& m, const State& s, const Prm& prm, int p) {
    double mn = 2.0 * SMGPU_PI, mx = 0.0;
    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
        const int e = m.peEdge[k];
        const double a = s.edgeMin[e], b = s.edgeMax[e];
        if (mn > a) mn = a;
        if (mx < b) mx = b;
    }
    s.ptMin[p] = mn;
    s.ptMax[p] = mx;
    const bool good = (mn > prm.smallAngle) && (mx < prm.largeAngle);
    noteNear(s, 1, nearTie(mn, prm.smallAngle, prm.nearUlps) + ((mn > prm.smallAngle) ? nearTie(mx, prm.largeAngle, prm.nearUlps) : 0));
    s.faActive[p] = good ? 0 : s.faGen;
    // one add per wave (the lanes that are here together), not one per point: 136 k same-address atomics per iteration on the
    // 10 M-cell cavity mesh otherwise
    const unsigned long long bad = __ballot(!good);
    if (bad && (int)(threadIdx.x & 63) == __ffsll((long long)bad) - 1) atomicAdd(&s.acc->nActive, __popcll(bad));
}
__global__ void __launch_bounds__(kBlock) k_fa_points(MeshView m, State s, Prm prm, const uint8_t* faMaybe) {
    if (s.acc->stop) return;
    if (faMaybe && s.acc->nFaMaybe == 0) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= m.nPoints) return;
    if (faMaybe && faMaybe[p] != s.faGen) return;   // all its edges are GOOD; its faActive mark is stale, i.e. clear
    faPointMinMax(m, s, prm, p);
}
__global__ void __launch_bounds__(kBlock) k_fa_points_list(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    const int n = s.acc->nFaPts;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) faPointMinMax(m, s, prm, s.faPointList[i]);
}

// calcMinMaxFaceAngleForPoint SM.C:1276-1308
__device__ __forceinline__ void pointFaceAngles(const MeshView& m, const State& s, int p, const V3& c1, int i2,
                                                const V3& c2, double& mn, double& mx) {
    mn = 2.0 * SMGPU_PI;
    mx = 0.0;
    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
        double a, b;
        edgeFaceAngles<true>(m, s, m.peEdge[k], p, c1, i2, c2, a, b);
        if (mn > a) mn = a;
        if (mx < b) mx = b;
    }
}

// Geometric predicates of the stack walk SM.C:1347-1434 for every point outside the good range.
// They are pure functions of (current coordinates, proposals), so they are evaluated in parallel
// here; k_fa_walk then replays the reference's stack order over these bits.
//   faS[p]  bit0: self move deteriorates (SM.C:1391-1394)   bit1: proposal differs from current
//   faN[k]  (k = pointPoints entry p->n)  bit0: n's move hurts p with p at its proposal
//           bit1: n's move hurts p with p at its current position   bit2: n is moving (SM.C:1414)
__global__ void __launch_bounds__(kBlock) k_fa_pred(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    if (s.acc->nActive == 0) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= m.nPoints) return;
    if (s.faActive[p] != s.faGen) return;
    const V3 cur = ldv(s.ptsCur, p);
    const V3 np = ldv(s.prop, p);
    const double curMin = s.ptMin[p], curMax = s.ptMax[p];
    auto bad = [&](double mn, double mx) {
        noteNear(s, 2, nearVerdict(mn, mx, curMin, curMax, prm));
        return ((mn < prm.smallAngle) && (mn < curMin)) || ((mx > prm.largeAngle) && (mx > curMax));
    };
    const bool moved = (np != cur);
    uint8_t sb = moved ? 2 : 0;
    if (moved) {
        double mn, mx;
        pointFaceAngles(m, s, p, np, -1, np, mn, mx);
        if (bad(mn, mx)) sb |= 1;
    }
    s.faS[p] = sb;
    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {
        const int q = m.ppPt[k];
        const V3 nq = ldv(s.prop, q);
        uint8_t nb = 0;
        if (nq != ldv(s.ptsCur, q)) {
            nb = 4;
            double mn, mx;
            pointFaceAngles(m, s, p, cur, q, nq, mn, mx);
            const bool badF = bad(mn, mx);
            if (badF) nb |= 2;
            if (moved) {
                pointFaceAngles(m, s, p, np, q, nq, mn, mx);
                if (bad(mn, mx)) nb |= 1;
            } else if (badF) nb |= 1;
        }
        s.faN[k] = nb;
    }
}

// Ordered replay of the stack walk SM.C:1347-1434: one wave; all lanes scan 64 flags at a time
// (ballot), lane 0 performs the sequential freeze logic.  Points are visited from the highest
// index down (the reference pushes 0..P-1 and pops from the top); a point frozen by a neighbour
// is re-visited before the walk continues (LIFO re-push, SM.C:1431).
__global__ void __launch_bounds__(64) k_fa_walk(MeshView m, State s) {
    if (s.acc->stop) return;
    if (s.acc->nActive == 0) return;
    const int lane = threadIdx.x;
    const int nChunks = (m.nPoints + 63) / 64;
    for (int c = nChunks - 1; c >= 0; --c) {
        const int idx = c * 64 + lane;
        const bool act = (idx < m.nPoints) && s.faActive[idx] == s.faGen;
        unsigned long long mask = __ballot(act);
        if (mask == 0ull) continue;
        if (lane == 0) {
            while (mask) {
                const int bit = 63 - __clzll((long long)mask);
                mask &= ~(1ull << bit);
                int sp = 0;
                s.walkStack[sp++] = c * 64 + bit;
                while (sp > 0) {
                    const int p = s.walkStack[--sp];
                    if (s.faActive[p] != s.faGen) continue;        // SM.C:1367-1369
                    const uint8_t sb = s.faS[p];
                    bool useNew = (!s.frozen[p]) && (sb & 2);      // SM.C:1376-1385
                    if (useNew && (sb & 1)) { s.frozen[p] = 1; useNew = false; }  // SM.C:1391-1399
                    for (int k = m.ppOff[p]; k < m.ppOff[p + 1]; ++k) {           // SM.C:1406-1433
                        const int q = m.ppPt[k];
                        const uint8_t nb = s.faN[k];
                        if (s.frozen[q]) continue;
                        if (!(nb & 4)) continue;
                        if (useNew ? (nb & 1) : (nb & 2)) { s.frozen[q] = 1; s.walkStack[sp++] = q; }
                    }
                }
            }
        }
    }
}

// restore + count SM.C:2384-2392, residual SM.C:1546-1565, movePoints SM.C:2399 -- one thread per point
__global__ void __launch_bounds__(kBlock) k_apply(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    double dist = 0.0;
    int fcount = 0;
    if (p < m.nPoints) {
        const uint8_t fl = m.pflags[p];
        const V3 cur = ldv(s.ptsCur, p);
        V3 np = ldv(s.prop, p);
        if (s.frozen[p] || (!(fl & PF_INTERNAL) && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
        dist = mag(np - cur) / prm.maxStep;
        stv(s.ptsNext, p, np);
    }
    blockPublish<kBlock>(s, dist, fcount, blockIdx.x);
}

// The same step when the proposal kernel has left |proposal - current|^2 per point (State::stepSqr): the proposal array BECOMES
// the next coordinates (the host swaps the two pointers after this launch), so only the points that do not move are written --
// 10 bytes per point read instead of k_apply's 50 read + 24 written.  Same values: the proposal kernel forms the square from
// the registers it stores the proposal from; sqrt and the division by maxStep are monotone, so the largest step is the root of
// the largest square over maxStep, bit for bit (one sqrt per thread instead of one per point); a restored point contributes 0
// here as mag(cur - cur) / maxStep does there; NaN never wins on either side.  Four consecutive points per thread.
constexpr int kApplyPer = 4;
__global__ void __launch_bounds__(kBlock) k_apply_swap(MeshView m, State s, Prm prm) {
    if (s.acc->stop) return;
    const int p0 = (blockIdx.x * kBlock + threadIdx.x) * kApplyPer;
    double sq[kApplyPer];
    unsigned fl[kApplyPer], fr[kApplyPer];
    const int n = min(kApplyPer, m.nPoints - p0);
    if (n == kApplyPer) {
        const unsigned f4 = *reinterpret_cast<const unsigned*>(m.pflags + p0), z4 = *reinterpret_cast<const unsigned*>(s.frozen + p0);
        const double2 a = *reinterpret_cast<const double2*>(s.stepSqr + p0), b = *reinterpret_cast<const double2*>(s.stepSqr + p0 + 2);
        sq[0] = a.x; sq[1] = a.y; sq[2] = b.x; sq[3] = b.y;
#pragma unroll
        for (int k = 0; k < kApplyPer; ++k) { fl[k] = (f4 >> (8 * k)) & 0xffu; fr[k] = (z4 >> (8 * k)) & 0xffu; }
    } else {
#pragma unroll
        for (int k = 0; k < kApplyPer; ++k) {
            const bool in = k < n;
            fl[k] = in ? m.pflags[p0 + k] : 0u; fr[k] = in ? s.frozen[p0 + k] : 0u; sq[k] = in ? s.stepSqr[p0 + k] : 0.0;
        }
    }
    double mx = 0.0;
    int fcount = 0;
#pragma unroll
    for (int k = 0; k < kApplyPer; ++k) {
        if (k >= n) break;
        if (fr[k] || (!(fl[k] & PF_INTERNAL) && !(fl[k] & PF_SMOOTHSURF))) { stv(s.prop, p0 + k, ldv(s.ptsCur, p0 + k)); ++fcount; }
        else if (sq[k] > mx) mx = sq[k];
    }
    blockPublish<kBlock>(s, sqrtExact(mx) / prm.maxStep, fcount, blockIdx.x);
}

// End of iteration: reduce the workgroup partials, publish the log-line values (SM.C:2396), stop test
// (SM.C:2401), reset accumulators.  One workgroup of 1024 threads, 4 independent loads in flight per thread.
constexpr int kFinishBlock = 1024;
constexpr int kStatsWritten = 1 << 30;
// the reduction by one workgroup of T threads (all of them call it)
template <int T>
__device__ __forceinline__ void finishPartials(const State& s, int nPartials, int iter, double relTol, double* localStats,
                                               double* history = nullptr) {
    Accum* a = s.acc;
    __shared__ double shMax[T / 64];
    __shared__ int shCnt[T / 64];
    double d = 0.0;
    int c = 0;
    for (int i0 = threadIdx.x; i0 < nPartials; i0 += 4 * T) {
        double v[4]; int n[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * T;
            v[u] = (i < nPartials) ? s.blkMax[i] : 0.0;
            n[u] = (i < nPartials) ? s.blkCnt[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { d = (v[u] > d) ? v[u] : d; c += n[u]; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double od = __shfl_down(d, o, 64);
        const int oc = __shfl_down(c, o, 64);
        d = (od > d) ? od : d;
        c += oc;
    }
    if ((threadIdx.x & 63) == 0) { shMax[threadIdx.x >> 6] = d; shCnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int i = 1; i < T / 64; ++i) { d = (shMax[i] > d) ? shMax[i] : d; c += shCnt[i]; }
    const double res = d;
    // (nNearTies on the device: bit 30 = "record written" for the host's read-back, the count below it)
    if (s.stats && iter >= 0) { s.stats[iter].residual = res; s.stats[iter].nFrozenPoints = c; s.stats[iter].nNearTies = kStatsWritten | (a->nNearTies < kStatsWritten ? a->nNearTies : kStatsWritten - 1); }
    if (localStats) { localStats[0] = res; localStats[1] = (double)c; }
    if (history) { history[0] = res; history[1] = (double)c; }
    if (res < relTol) a->stop = 1;
    if (s.nActiveHost) __hip_atomic_store(s.nActiveHost, a->nActive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a->nActive = 0;
    a->nEaMaybe = 0;
    a->nFaMaybe = 0;
    a->nFaEdges = 0;
    a->nFaPts = 0;
    a->nNearTies = 0;
}
__global__ void __launch_bounds__(kFinishBlock) k_finish(State s, int nPartials, int iter, double relTol, double* localStats,
                                                          double* history) {
    if (s.acc->stop) return;
    finishPartials<kFinishBlock>(s, nPartials, iter, relTol, localStats, history);
}

// ---- multi-rank pack / combine -------------------------------------------------------------------
// exchange A: local partial sums and closest points of the shared points (SM.C:108-131, 325-387)
__global__ void __launch_bounds__(kBlock) k_halo_packA(MeshView m, State s, const int* sharedLocal, double* ownA, int nShared,
                                                        const int* sendOff, const int* sendSlots, double* sendA, int centroidAll) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    // the own record (one per shared point) and its copies in the send slots of the point (one per other sharer)
    if (i < nShared) {
        const int p = sharedLocal[i];
        const V3 cur = ldv(s.ptsCur, p);
        PointLocal L;
        int err = 0;
        pointLocal(m, s, p, cur, m.pflags[p] & PF_INTERNAL, L, err, centroidAll != 0);
        if (err) s.acc->err = 1;
        double* r = ownA + (size_t)i * SMGPU_HALO_A_DOUBLES;
        r[0] = L.sum.x; r[1] = L.sum.y; r[2] = L.sum.z;
        r[3] = L.r1.x; r[4] = L.r1.y; r[5] = L.r1.z;
        r[6] = L.r2.x; r[7] = L.r2.y; r[8] = L.r2.z;
        r[9] = L.r3.x; r[10] = L.r3.y; r[11] = L.r3.z;
        r[12] = __longlong_as_double(((long long)L.hc << 32) | (long long)(unsigned int)L.count);
        for (int k = sendOff[i]; k < sendOff[i + 1]; ++k) {
            double* q = sendA + (size_t)sendSlots[k] * SMGPU_HALO_A_DOUBLES;
#pragma unroll
            for (int j = 0; j < SMGPU_HALO_A_DOUBLES; ++j) q[j] = r[j];
        }
    }
}

// boundary layer treatment / boundary point smoothing under -parallel: the record of SMGPU_HALO_L_DOUBLES per shared point
// (layout in include/smgpu.h): local normal and outer neighbour coordinates (UNDEF_VECTOR when the neighbour is not in this
// rank, OBB.C:474-478); with boundary point smoothing also the local face count, the inner neighbour's coordinates and the
// local feature edge projections (bf* / inner / feat* = the boundary tables of kernels_boundary.hpp, or null).
struct PackLArgs {
    const int* sharedLocal; double* ownL; int nShared; const int* sendOff; const int* sendSlots; double* sendL;
    const int* bfOff; const int* inner; const int* featOfBnd; const double* featSum; const int* featCnt;
};
__device__ __forceinline__ void haloPackLOf(const State& s, const PackLArgs& a, int i) {
    const int* sharedLocal = a.sharedLocal; double* ownL = a.ownL; const int nShared = a.nShared;
    const int *sendOff = a.sendOff, *sendSlots = a.sendSlots; double* sendL = a.sendL;
    const int *bfOff = a.bfOff, *inner = a.inner, *featOfBnd = a.featOfBnd; const double* featSum = a.featSum; const int* featCnt = a.featCnt;
    if (i >= nShared) return;
    const int p = sharedLocal[i];
    const int q = s.layerMap ? s.layerMap[p] : -1;
    double rec[SMGPU_HALO_L_DOUBLES];
    const V3 nrm = ldv(s.layerNormal, p);
    const V3 x = q >= 0 ? ldv(s.ptsCur, q) : v3(SMGPU_GREAT, SMGPU_GREAT, SMGPU_GREAT);
    rec[0] = nrm.x; rec[1] = nrm.y; rec[2] = nrm.z; rec[3] = x.x; rec[4] = x.y; rec[5] = x.z;
    rec[6] = 0.0; rec[7] = rec[8] = rec[9] = SMGPU_GREAT; rec[10] = rec[11] = rec[12] = 0.0; rec[13] = 0.0;
    const int bi = s.bndOfShared ? s.bndOfShared[i] : -1;
    if (bi >= 0) {
        rec[6] = (double)(bfOff[bi + 1] - bfOff[bi]);
        if (inner[bi] >= 0) { const V3 y = ldv(s.ptsCur, inner[bi]); rec[7] = y.x; rec[8] = y.y; rec[9] = y.z; }   // OBB.C:471-486
        const int f = featOfBnd[bi];
        if (f >= 0) { const V3 fs = ldv(featSum, f); rec[10] = fs.x; rec[11] = fs.y; rec[12] = fs.z; rec[13] = (double)featCnt[f]; }
    }
    const int w = s.lStride;
    double* o = ownL + (size_t)i * w;
    for (int j = 0; j < w; ++j) o[j] = rec[j];
    for (int k = sendOff[i]; k < sendOff[i + 1]; ++k) {
        if (s.push.slotL) {   // (records of 6 or 14 doubles: 16-byte aligned)
            double* d = s.push.slotL[sendSlots[k]];
            if (w == SMGPU_HALO_L_LAYERS) { stPeer2(d, rec[0], rec[1]); stPeer2(d + 2, rec[2], rec[3]); stPeer2(d + 4, rec[4], rec[5]); }
            else {
#pragma unroll
                for (int j = 0; j < SMGPU_HALO_L_DOUBLES; j += 2) stPeer2(d + j, rec[j], rec[j + 1]);
            }
            continue;
        }
        double* d = sendL + (size_t)sendSlots[k] * w;
        for (int j = 0; j < w; ++j) d[j] = rec[j];
    }
}
__global__ void __launch_bounds__(kBlock) k_halo_packL(State s, PackLArgs a) { haloPackLOf(s, a, blockIdx.x * kBlock + threadIdx.x); }
// plusEq in ascending rank order for the normals, face counts and feature projections (OBB.C:184-198, BPS.C:659-674);
// minMagSqrEqOp for the outer and inner neighbour coordinates (OBB.C:490-496): the master's fold -- the lowest rank's value, the
// others folded onto it in ascending rank order, the same result on every sharer (globalMeshData::syncData) -- or, ownFold, the
// others folded onto the own value
__device__ __forceinline__ void haloCombineLOf(int i, int nShared, const int* combOff, const int* combSlots, const double* ownL,
                                               const double* recvL, double* combL, int w, int ownFold) {
    if (i >= nShared) return;
    const int b = combOff[i], n = combOff[i + 1] - b;
    const double* own = ownL + (size_t)i * w;
    const bool wide = w > SMGPU_HALO_L_LAYERS;   // the boundary point smoothing fields travel too
    V3 sum = v3(0, 0, 0), fsum = v3(0, 0, 0);
    double faces = 0.0, fcnt = 0.0;
    V3 x = v3(own[3], own[4], own[5]), y = v3(wide ? own[7] : SMGPU_GREAT, wide ? own[8] : SMGPU_GREAT, wide ? own[9] : SMGPU_GREAT);
    for (int j = 0; j < n; ++j) {
        const int sl = combSlots[b + j];
        const double* r = (sl < 0) ? own : recvL + (size_t)sl * w;
        sum = sum + v3(r[0], r[1], r[2]);
        if (wide) {
            faces += r[6];
            fsum = fsum + v3(r[10], r[11], r[12]);
            fcnt += r[13];
        }
        const V3 x2 = v3(r[3], r[4], r[5]), y2 = v3(wide ? r[7] : SMGPU_GREAT, wide ? r[8] : SMGPU_GREAT, wide ? r[9] : SMGPU_GREAT);
        if (!ownFold && j == 0) { x = x2; y = y2; }        // the master's value starts the fold
        else if (!ownFold || sl >= 0) {
            { const bool k = magSqr(x) <= magSqr(x2); x = v3(k ? x.x : x2.x, k ? x.y : x2.y, k ? x.z : x2.z); }
            if (wide) { const bool k = magSqr(y) <= magSqr(y2); y = v3(k ? y.x : y2.x, k ? y.y : y2.y, k ? y.z : y2.z); }
        }
    }
    double* o = combL + (size_t)i * w;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z; o[3] = x.x; o[4] = x.y; o[5] = x.z;
    if (wide) { o[6] = faces; o[7] = y.x; o[8] = y.y; o[9] = y.z; o[10] = fsum.x; o[11] = fsum.y; o[12] = fsum.z; o[13] = fcnt; }
}
__global__ void __launch_bounds__(kBlock, 2) k_halo_combineL(int nShared, const int* combOff, const int* combSlots, const double* ownL,
                                                          const double* recvL, double* combL, int w, int ownFold) {
    haloCombineLOf(blockIdx.x * kBlock + threadIdx.x, nShared, combOff, combSlots, ownL, recvL, combL, w, ownFold);
}

// SM.C:246-272 isCloserPoint
__device__ __forceinline__ bool isCloserPoint(const V3& a, const V3& b) {
    if (a == b) return false;
    const double delta = mag(a) - mag(b);
    if (delta < SMGPU_VSMALL) return true;
    return false;  // the |delta| < VSMALL branch (SM.C:266) cannot be reached once the test above failed
}

// The three sequential syncs (SM.C:391-478) for a point with TWO sharers, both ranks' views in scalars: ra = this rank's
// record, rb = the other rank's, selfFirst = this rank is the lower one (plusEqOp sums in ascending rank order).
// minMagSqrEqOp (x = (magSqr(x) <= magSqr(y)) ? x : y): syncTools::syncPointList folds once, master (lower rank) first, and both
// ranks receive that value -- an exact tie hands the LOWER rank's vector to both, which is what lets isCloserPoint's equal-length
// case fire on the upper rank; ownFold = each rank folds the other's value onto its own (a tie keeps the own vector on both).
__device__ __forceinline__ void combineTwoSharers(const double* ra, const double* rb, bool selfFirst, int ownFold, V3& sum, V3& a1, V3& a2, V3& a3,
                                                  int& cnt, int& anyCommon) {
    const V3 sa = v3(ra[0], ra[1], ra[2]), sb = v3(rb[0], rb[1], rb[2]);
    sum = selfFirst ? (v3(0, 0, 0) + sa) + sb : (v3(0, 0, 0) + sb) + sa;    // plusEqOp, ascending rank
    a1 = v3(ra[3], ra[4], ra[5]); a2 = v3(ra[6], ra[7], ra[8]); a3 = v3(ra[9], ra[10], ra[11]);
    V3 b1 = v3(rb[3], rb[4], rb[5]), b2 = v3(rb[6], rb[7], rb[8]), b3 = v3(rb[9], rb[10], rb[11]);
    const long long pa = __double_as_longlong(ra[12]), pb = __double_as_longlong(rb[12]);
    cnt = (int)(pa & 0xffffffffll) + (int)(pb & 0xffffffffll);
    int hcA = (int)(pa >> 32), hcB = (int)(pb >> 32);
    const bool aLeads = ownFold || selfFirst, bLeads = ownFold || !selfFirst;   // whose value starts the fold each rank receives
    // (selects by component: `c ? X : Y` on two V3 objects is a select of their ADDRESSES, which keeps both in scratch memory)
    auto pick = [](bool c, const V3& x, const V3& y) { return v3(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z); };
#define SMGPU_FOLD2(X, Y) pick(magSqr(X) <= magSqr(Y), (X), (Y))
    {   // SM.C:397-419: the first vectors
        const V3 fab = SMGPU_FOLD2(a1, b1), fba = SMGPU_FOLD2(b1, a1);
        const V3 svA = pick(aLeads, fab, fba), svB = pick(bLeads, fba, fab);
        if (isCloserPoint(svA, a1)) { a3 = a2; a2 = a1; a1 = svA; hcA = 0; }
        if (isCloserPoint(svB, b1)) { b3 = b2; b2 = b1; b1 = svB; hcB = 0; }
    }
    {   // SM.C:424-445: the (updated) second vectors
        const V3 fab = SMGPU_FOLD2(a2, b2), fba = SMGPU_FOLD2(b2, a2);
        const V3 svA = pick(aLeads, fab, fba), svB = pick(bLeads, fba, fab);
        if (isCloserPoint(svA, a2)) { a3 = a2; a2 = svA; hcA = 0; }
        if (isCloserPoint(svB, b2)) { b3 = b2; b2 = svB; hcB = 0; }
    }
    {   // SM.C:450-469: the (updated) third vectors
        const V3 svA = pick(aLeads, SMGPU_FOLD2(a3, b3), SMGPU_FOLD2(b3, a3));
        if (isCloserPoint(svA, a3)) a3 = svA;
    }
#undef SMGPU_FOLD2
    anyCommon = hcA | hcB;
}

// The same three syncs written for few live registers (k_smooth_halo combines inside the smoothing launch, where the kernel's
// register count is the maximum over all roles): what a rank's state needs from an earlier step is a decision bit, and the
// vectors come from memory again when a step asks for them -- at most two vectors of each rank are live at a time, every final
// value is stored as soon as it stands.  Same operations on the same operands as combineTwoSharers, hence the same bits
// (tests/test_gpu_multirank.py runs both forms against the oracle, the tie cases included).
__device__ __forceinline__ void combineTwoToMemory(const double* __restrict__ ra, const double* __restrict__ rb, bool selfFirst, int ownFold,
                                                   double* __restrict__ o) {
    // (the received record with system-scope loads: with the flagged arrangement it was written by the host's exchange kernel
    // while this launch was already running)
    // The whole received record in ONE round trip, up front (13 independent loads: 26 registers, which this place can afford --
    // nothing else is live yet); the own record comes from the local L2 as the steps ask for it.
    auto ld3 = [](const double* p) { return v3(p[0], p[1], p[2]); };
    auto pick = [](bool c, const V3& x, const V3& y) { return v3(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z); };
    double rbv[SMGPU_HALO_A_DOUBLES];
#pragma unroll
    for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) rbv[q] = ldSys(rb + q);
    {   // plusEqOp, ascending rank
        const V3 sa = ld3(ra), sb = v3(rbv[0], rbv[1], rbv[2]);
        const V3 sum = selfFirst ? (v3(0, 0, 0) + sa) + sb : (v3(0, 0, 0) + sb) + sa;
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
    }
    const bool aLeads = ownFold || selfFirst, bLeads = ownFold || !selfFirst;
#define SMGPU_FOLD2(X, Y) pick(magSqr(X) <= magSqr(Y), (X), (Y))
    bool dA1, dB1, dA2, dB2;
    V3 a2, b2;      // the ranks' (updated) second vectors
    {   // SM.C:397-419
        const V3 a1 = ld3(ra + 3), b1 = v3(rbv[3], rbv[4], rbv[5]);
        const V3 fab = SMGPU_FOLD2(a1, b1), fba = SMGPU_FOLD2(b1, a1);
        const V3 svA = pick(aLeads, fab, fba), svB = pick(bLeads, fba, fab);
        dA1 = isCloserPoint(svA, a1); dB1 = isCloserPoint(svB, b1);
        const V3 f1 = pick(dA1, svA, a1);
        o[3] = f1.x; o[4] = f1.y; o[5] = f1.z;
        a2 = pick(dA1, a1, ld3(ra + 6));
        b2 = pick(dB1, b1, v3(rbv[6], rbv[7], rbv[8]));
    }
    V3 a3, b3;      // ... third vectors
    {   // SM.C:424-445
        const V3 fab = SMGPU_FOLD2(a2, b2), fba = SMGPU_FOLD2(b2, a2);
        const V3 svA = pick(aLeads, fab, fba), svB = pick(bLeads, fba, fab);
        dA2 = isCloserPoint(svA, a2); dB2 = isCloserPoint(svB, b2);
        const V3 f2 = pick(dA2, svA, a2);
        o[6] = f2.x; o[7] = f2.y; o[8] = f2.z;
        // the third vector before this step: the original second one if the first step shifted, else the original third one
        a3 = pick(dA2, a2, ld3(ra + (dA1 ? 6 : 9)));
        b3 = pick(dB2, b2, pick(dB1, v3(rbv[6], rbv[7], rbv[8]), v3(rbv[9], rbv[10], rbv[11])));
    }
    {   // SM.C:450-469
        const V3 svA = pick(aLeads, SMGPU_FOLD2(a3, b3), SMGPU_FOLD2(b3, a3));
        const V3 f3 = pick(isCloserPoint(svA, a3), svA, a3);
        o[9] = f3.x; o[10] = f3.y; o[11] = f3.z;
    }
#undef SMGPU_FOLD2
    const long long pa = __double_as_longlong(ra[12]), pb = __double_as_longlong(rbv[12]);
    const int cnt = (int)(pa & 0xffffffffll) + (int)(pb & 0xffffffffll);
    const int hcA = (dA1 || dA2) ? 0 : (int)(pa >> 32), hcB = (dB1 || dB2) ? 0 : (int)(pb >> 32);
    o[12] = __longlong_as_double(((long long)(hcA | hcB) << 32) | (long long)(unsigned int)cnt);
}

constexpr int kMaxSharers = 16;

// k_halo_combineA for the usual case (no point with more than 16 sharers): the two-sharer points from a per-point table of
// the other rank's receive slot (bit 30: this rank is the lower one; -1: not a two-sharer point) -- offset -> slot -> record
// was three dependent loads, this is two -- and the points with more sharers in the trailing workgroups.  No per-sharer
// arrays here, so the kernel needs no scratch memory.
template <bool COH = false>
__device__ __forceinline__ void combineMulti(int blk, int nMulti, const int* multiIdx, const int* multiSlots,
                                             const double* ownA, const double* recvA, double* combA, int ownFold);
__device__ __forceinline__ void haloCombineA2Of(int bx, int nShared, const int* __restrict__ peer, const double* __restrict__ ownA,
                                                const double* __restrict__ recvA, double* __restrict__ combA, int nBlocksTwo, int nMulti,
                                                const int* multiIdx, const int* multiSlots, int ownFold) {
    if (bx >= nBlocksTwo) {
        combineMulti(bx - nBlocksTwo, nMulti, multiIdx, multiSlots, ownA, recvA, combA, ownFold);
        return;
    }
    const int i = bx * kBlock + threadIdx.x;
    if (i >= nShared) return;
    const int pr = peer[i];
    if (pr < 0) return;
    const double* ra = ownA + (size_t)i * SMGPU_HALO_A_DOUBLES;
    const double* rb = recvA + (size_t)(pr & 0x3fffffff) * SMGPU_HALO_A_DOUBLES;
    V3 sum, a1, a2, a3;
    int cnt, any2;
    combineTwoSharers(ra, rb, (pr & 0x40000000) != 0, ownFold, sum, a1, a2, a3, cnt, any2);
    double* o = combA + (size_t)i * SMGPU_HALO_A_DOUBLES;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
    o[3] = a1.x; o[4] = a1.y; o[5] = a1.z;
    o[6] = a2.x; o[7] = a2.y; o[8] = a2.z;
    o[9] = a3.x; o[10] = a3.y; o[11] = a3.z;
    o[12] = __longlong_as_double(((long long)any2 << 32) | (long long)(unsigned int)cnt);
}
// (launch bounds: two workgroups per CU suffice for these latency-bound launches; the default occupancy target capped the
// registers at 80 and spilled the two ranks' vectors)
__global__ void __launch_bounds__(kBlock, 2) k_halo_combineA2(int nShared, const int* __restrict__ peer, const double* __restrict__ ownA,
                                                           const double* __restrict__ recvA, double* __restrict__ combA, int nBlocksTwo, int nMulti,
                                                           const int* multiIdx, const int* multiSlots, PushWait pw, int ownFold) {
    pushWait(pw);
    haloCombineA2Of((int)blockIdx.x, nShared, peer, ownA, recvA, combA, nBlocksTwo, nMulti, multiIdx, multiSlots, ownFold);
}

// syncPointList semantics for one shared point (same model as oracle MultiDomain::syncA):
// plusEqOp in ascending rank order; minMagSqrEqOp folds from the master's (lowest rank's) value in ascending rank order and every
// sharer receives that result (ties keep the lower rank's vector) -- or, ownFold, from each sharer's own value.
// Points with more than two sharers (processor edges and corners: few) are left to combineMulti when skipMulti
// is set: their per-sharer arrays live in scratch memory and a single lane walking them cost ~70 us per launch.
// The workgroups after the first nBlocksTwo handle the listed points with more than two sharers (combineMulti): one launch,
// so that the latency of that small, dependent-load-bound part overlaps with the two-sharer part.
__global__ void __launch_bounds__(kBlock, 2) k_halo_combineA(int nShared, const int* combOff, const int* combSlots,
                                                          const double* ownA, const double* recvA, double* combA, int* err,
                                                          int skipMulti, int nBlocksTwo, int nMulti, const int* multiIdx,
                                                          const int* multiSlots, PushWait pw, int ownFold) {
    pushWait(pw);
    if ((int)blockIdx.x >= nBlocksTwo) {
        combineMulti((int)blockIdx.x - nBlocksTwo, nMulti, multiIdx, multiSlots, ownA, recvA, combA, ownFold);
        return;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nShared) return;
    const int b = combOff[i], n = combOff[i + 1] - b;
    if (n > kMaxSharers) { *err = 2; return; }
    if (n > 2 && skipMulti) return;
    if (n == 2) {
        // two sharers (a point inside a processor face: nearly all shared points): the same three sequential syncs with
        // both ranks' views in scalars -- the general form below keeps per-sharer arrays, which live in scratch memory
        const int s0 = combSlots[b], s1 = combSlots[b + 1];
        const double* ra = ownA + (size_t)i * SMGPU_HALO_A_DOUBLES;                       // A = this rank
        const double* rb = recvA + (size_t)(s0 < 0 ? s1 : s0) * SMGPU_HALO_A_DOUBLES;     // B = the other one
        const bool selfFirst = s0 < 0;
        V3 sum, a1, a2, a3;
        int cnt, any2;
        combineTwoSharers(ra, rb, selfFirst, ownFold, sum, a1, a2, a3, cnt, any2);
        double* o = combA + (size_t)i * SMGPU_HALO_A_DOUBLES;
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
        o[3] = a1.x; o[4] = a1.y; o[5] = a1.z;
        o[6] = a2.x; o[7] = a2.y; o[8] = a2.z;
        o[9] = a3.x; o[10] = a3.y; o[11] = a3.z;
        o[12] = __longlong_as_double(((long long)any2 << 32) | (long long)(unsigned int)cnt);
        return;
    }
    V3 r1[kMaxSharers], r2[kMaxSharers], r3[kMaxSharers];
    int hc[kMaxSharers];
    V3 sum = v3(0, 0, 0);
    int cnt = 0, self = 0;
    for (int j = 0; j < n; ++j) {
        const int sl = combSlots[b + j];
        const double* r = (sl < 0) ? ownA + (size_t)i * SMGPU_HALO_A_DOUBLES : recvA + (size_t)sl * SMGPU_HALO_A_DOUBLES;
        if (sl < 0) self = j;
        sum = sum + v3(r[0], r[1], r[2]);
        r1[j] = v3(r[3], r[4], r[5]);
        r2[j] = v3(r[6], r[7], r[8]);
        r3[j] = v3(r[9], r[10], r[11]);
        const long long pk = __double_as_longlong(r[12]);
        cnt += (int)(pk & 0xffffffffll);
        hc[j] = (int)(pk >> 32);
    }
    auto fold = [&](const V3* v, int me) {
        const int lead = ownFold ? me : 0;         // whose value starts the fold: the own one, or the master's (lowest rank)
        V3 x = v[lead];
        for (int k = 0; k < n; ++k) {
            if (k == lead) continue;
            x = (magSqr(x) <= magSqr(v[k])) ? x : v[k];
        }
        return x;
    };
    V3 sent[kMaxSharers];
    for (int j = 0; j < n; ++j) sent[j] = r1[j];
    for (int j = 0; j < n; ++j) {            // SM.C:397-419
        const V3 sv = fold(sent, j);
        if (isCloserPoint(sv, r1[j])) { r3[j] = r2[j]; r2[j] = r1[j]; r1[j] = sv; hc[j] = 0; }
    }
    for (int j = 0; j < n; ++j) sent[j] = r2[j];
    for (int j = 0; j < n; ++j) {            // SM.C:424-445
        const V3 sv = fold(sent, j);
        if (isCloserPoint(sv, r2[j])) { r3[j] = r2[j]; r2[j] = sv; hc[j] = 0; }
    }
    for (int j = 0; j < n; ++j) sent[j] = r3[j];
    for (int j = 0; j < n; ++j) {            // SM.C:450-469
        const V3 sv = fold(sent, j);
        if (isCloserPoint(sv, r3[j])) r3[j] = sv;
    }
    int any = 0;                             // SM.C:472-478
    for (int j = 0; j < n; ++j) any |= hc[j];
    double* o = combA + (size_t)i * SMGPU_HALO_A_DOUBLES;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
    o[3] = r1[self].x; o[4] = r1[self].y; o[5] = r1[self].z;
    o[6] = r2[self].x; o[7] = r2[self].y; o[8] = r2[self].z;
    o[9] = r3[self].x; o[10] = r3[self].y; o[11] = r3[self].z;
    o[12] = __longlong_as_double(((long long)any << 32) | (long long)(unsigned int)cnt);
}

// The same three syncs for the shared points with 3..16 sharers, 16 lanes per point: lane j plays sharer j (ascending
// rank), holds its three vectors in registers and reads the others' values through LDS (wave shuffles of doubles were
// measured ~4x slower here).  multiIdx lists those points.
// multiSlots: 16 entries per listed point -- the recv slot of sharer j, -1 = this rank, -2 = no such sharer (one coalesced
// load instead of the multiIdx -> combOff -> combSlots chain: this part is a handful of workgroups and latency bound)
// COH: the combined record is read by other workgroups of the SAME launch (k_smooth_halo): coherent stores
template <bool COH>
__device__ __forceinline__ void combineMulti(int blk, int nMulti, const int* multiIdx, const int* multiSlots,
                                             const double* ownA, const double* recvA, double* combA, int ownFold) {
    // 16 lanes per point, lane j = sharer j (ascending rank); the sharers' values travel by shuffles within the group -- no
    // LDS, no workgroup barriers (the LDS form spent most of the launch in its ten barriers)
    const int t = threadIdx.x, g = (blk * kBlock + t) >> 4, j = t & 15, base = t & 48;      // base: the group's first lane in the wave
    const bool live = g < nMulti;
    const int sl = live ? multiSlots[g * 16 + j] : -2;
    const int i = live ? multiIdx[g] : 0;
    const bool mine = sl > -2;
    const unsigned long long bal = __ballot(mine);
    const int n = __popcll((bal >> base) & 0xffffull);      // sharers occupy the first n lanes of the group
    const double* r = (mine && sl < 0) ? ownA + (size_t)i * SMGPU_HALO_A_DOUBLES : recvA + (size_t)(mine ? sl : 0) * SMGPU_HALO_A_DOUBLES;
    V3 sv0 = v3(0, 0, 0), r1 = sv0, r2 = sv0, r3 = sv0;
    int cntJ = 0, hc = 0;
    if (mine) {
        double w[SMGPU_HALO_A_DOUBLES];
#pragma unroll
        for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) w[q] = (COH && sl >= 0) ? ldSys(r + q) : r[q];      // (see combineTwoToMemory)
        sv0 = v3(w[0], w[1], w[2]);
        r1 = v3(w[3], w[4], w[5]); r2 = v3(w[6], w[7], w[8]); r3 = v3(w[9], w[10], w[11]);
        const long long pk = __double_as_longlong(w[12]);
        cntJ = (int)(pk & 0xffffffffll);
        hc = (int)(pk >> 32);
    }
#define SMGPU_FROM(V, K) v3(__shfl((V).x, base + (K), 64), __shfl((V).y, base + (K), 64), __shfl((V).z, base + (K), 64))
    // plusEqOp in ascending rank order (every lane forms the same sum), count
    V3 sum = v3(0, 0, 0);
    int cnt = 0;
    for (int k = 0; k < n; ++k) { sum = sum + SMGPU_FROM(sv0, k); cnt += __shfl(cntJ, base + k, 64); }
    // minMagSqrEqOp folded from the master's value (lane 0 of the group: the lowest rank) over the others in ascending rank
    // order, the same result on every lane -- or, ownFold, from the lane's own value
    const int lead = ownFold ? j : 0;
#define SMGPU_FOLD_ALL(SENT, OUT)                                                          \
    {                                                                                      \
        const V3 sent_ = (SENT);                                                           \
        const V3 m0_ = SMGPU_FROM(sent_, 0);                                               \
        V3 x_ = v3(ownFold ? sent_.x : m0_.x, ownFold ? sent_.y : m0_.y, ownFold ? sent_.z : m0_.z); \
        for (int k = 0; k < n; ++k) {                                                      \
            const V3 y_ = SMGPU_FROM(sent_, k);                                            \
            if (k != lead) { const bool kp_ = magSqr(x_) <= magSqr(y_); x_ = v3(kp_ ? x_.x : y_.x, kp_ ? x_.y : y_.y, kp_ ? x_.z : y_.z); } \
        }                                                                                  \
        (OUT) = x_;                                                                        \
    }
    V3 sv;
    SMGPU_FOLD_ALL(r1, sv)                       // SM.C:397-419
    if (mine && isCloserPoint(sv, r1)) { r3 = r2; r2 = r1; r1 = sv; hc = 0; }
    SMGPU_FOLD_ALL(r2, sv)                       // SM.C:424-445
    if (mine && isCloserPoint(sv, r2)) { r3 = r2; r2 = sv; hc = 0; }
    SMGPU_FOLD_ALL(r3, sv)                       // SM.C:450-469
    if (mine && isCloserPoint(sv, r3)) r3 = sv;
#undef SMGPU_FOLD_ALL
#undef SMGPU_FROM
    int any = 0;                                 // SM.C:472-478
    for (int k = 0; k < n; ++k) any |= __shfl(hc, base + k, 64);
    if (mine && sl < 0) {
        double* o = combA + (size_t)i * SMGPU_HALO_A_DOUBLES;
        const double rec[SMGPU_HALO_A_DOUBLES] = {sum.x, sum.y, sum.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z, r3.x, r3.y, r3.z,
                                                  __longlong_as_double(((long long)any << 32) | (long long)(unsigned int)cnt)};
#pragma unroll
        for (int q = 0; q < SMGPU_HALO_A_DOUBLES; ++q) { if (COH) stCoh(o + q, rec[q]); else o[q] = rec[q]; }
    }
}

// exchange F: isFrozenPoint of shared points, orEqOp (SM.C:2374-2380)
__global__ void __launch_bounds__(kBlock) k_halo_packF(int nSend, const int* sendShared, const int* sharedLocal,
                                                       const uint8_t* frozen, int* sendF, PushView pv, unsigned tag) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nSend) {
        const int v = frozen[sharedLocal[sendShared[i]]];
        if (pv.slotF) stPeer(pv.slotF[i], v); else sendF[i] = v;
    }
    pushSignal(pv, 1, tag);
}
__global__ void __launch_bounds__(kBlock) k_halo_orF(int nShared, const int* sharedLocal, const int* combOff,
                                                     const int* combSlots, const int* recvF, uint8_t* frozen, PushWait pw) {
    pushWait(pw);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nShared) return;
    int any = 0;
    for (int k = combOff[i]; k < combOff[i + 1]; ++k) {
        const int sl = combSlots[k];
        if (sl >= 0) any |= recvF[sl];
    }
    if (any) frozen[sharedLocal[i]] = 1;
}

// Multi-rank, constraints off: the fused proposal kernel has already moved every non-shared point; for
// the shared points it left the proposal and the local freeze flag.  After exchange F this kernel ORs the
// flags (SM.C:2374), restores / counts (SM.C:2384-2392), writes the new coordinates and publishes the
// residual partials of the shared points (partial slots after the tile slots).
__global__ void __launch_bounds__(kBlock) k_shared_fix(MeshView m, State s, Prm prm, int nShared, const int* sharedLocal,
                                                       const int* combOff, const int* combSlots, const int* recvF, int partialBase, PushWait pw, int sysLoads = 0) {
    if (s.acc->stop) return;
    pushWait(pw);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double dist = 0.0;
    int fcount = 0;
    if (i < nShared) {
        const int p = sharedLocal[i];
        int frz = s.frozen[p];
        for (int k = combOff[i]; k < combOff[i + 1]; ++k) {
            const int sl = combSlots[k];
            if (sl >= 0) frz |= sysLoads ? ldSys(recvF + sl) : recvF[sl];
        }
        s.frozen[p] = frz ? 1 : 0;
        const uint8_t fl = m.pflags[p];
        const V3 cur = ldv(s.ptsCur, p);
        V3 np = ldv(s.prop, p);
        if (frz || (!(fl & PF_INTERNAL) && !(fl & PF_SMOOTHSURF))) { np = cur; fcount = 1; }
        dist = mag(np - cur) / prm.maxStep;
        stv(s.ptsNext, p, np);
    }
    blockPublish<kBlock>(s, dist, fcount, partialBase + blockIdx.x);
}

// getMeshStats SM.C:1495-1510: min / max edge length.  Grid-stride loop, one pair of atomics per workgroup (a few thousand
// atomics on one word serialise at ~90 per microsecond).
__global__ void __launch_bounds__(kBlock) k_edge_stats(MeshView m, const double* pts, unsigned long long* minBits,
                                                       unsigned long long* maxBits) {
    __shared__ double shMin[kBlock / 64], shMax[kBlock / 64];
    double mn = 1.0e300, mx = 0.0;
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < m.nEdges; e += gridDim.x * kBlock) {
        const double len = mag(ldv(pts, m.edges[2 * e + 1]) - ldv(pts, m.edges[2 * e]));
        mn = (len < mn) ? len : mn;
        mx = (len > mx) ? len : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_down(mn, o, 64), b = __shfl_down(mx, o, 64);
        mn = (a < mn) ? a : mn;
        mx = (b > mx) ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) { shMin[threadIdx.x >> 6] = mn; shMax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) { mn = (shMin[w] < mn) ? shMin[w] : mn; mx = (shMax[w] > mx) ? shMax[w] : mx; }
        atomicMin(minBits, (unsigned long long)__double_as_longlong(mn));
        atomicMax(maxBits, (unsigned long long)__double_as_longlong(mx));
    }
}

}  // namespace smgpu
