// smgpu.hip -- C-ABI implementation (see include/smgpu.h): host-side addressing build, device
// residency, kernel sequencing of one smoothing iteration (src/smoothMesh.C:2257-2437).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <future>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "layers.hpp"
#include "kernels_tiled.hpp"
#include "kernels_walk.hpp"
#include "kernels_filter.hpp"
#include "kernels_boundary.hpp"
#include "boundary.hpp"
#include "tiles.hpp"
#include "topology.hpp"
#include "parallel.hpp"

using namespace smgpu;

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" __FILE__ ":" +       \
                        std::to_string(__LINE__) + ")");                                               \
    } while (0)

enum KernelId { K_FACE_GEOM = 0, K_CELL_CENTRES, K_SMOOTH_FINAL, K_SMOOTH_PROP, K_EDGE_ANGLE, K_FA_EDGES,
                K_FA_POINTS, K_FA_PRED, K_FA_WALK, K_APPLY, K_FINISH, K_HALO, K_GEOM_TILE, K_EA_FILTER, K_FA_FILTER, K_BND, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"k_face_geom", "k_cell_centres", "k_smooth<final>", "k_smooth<proposal>",
                                            "k_edge_angle", "k_fa_edges", "k_fa_points", "k_fa_pred", "k_fa_walk",
                                            "k_apply", "k_finish", "k_halo_*", "k_geom_tile", "k_edge_angle_filter", "k_fa_edges_filter", "k_bnd_*"};

struct smgpu_handle {
    Topology topo;
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::vector<void*> allocs;
    int64_t deviceBytes = 0;
    MeshView mv{};
    State st{};
    double* bufA = nullptr;  // coordinate double buffer
    double* bufB = nullptr;
    smgpu_params prm{};
    bool prmSet = false;
    smgpu_iter_stats* dStats = nullptr;
    int statsCap = 0;
    // timing
    bool timing = false;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> freeEvents;
    double ms[K_COUNT] = {0};
    int64_t launches[K_COUNT] = {0};
    int64_t algoBytes[K_COUNT] = {0};
    int64_t algoF64[K_COUNT] = {0};
    // halo
    bool haloOn = false;
    int nShared = 0, nSend = 0, nRecv = 0;
    int *dSharedLocal = nullptr, *dSendShared = nullptr, *dCombOff = nullptr, *dCombSlots = nullptr, *dSharedSlot = nullptr;
    int *dSendOff = nullptr, *dSendSlots = nullptr;   // CSR: shared point -> its send slots
    double *dOwnA = nullptr, *dCombA = nullptr;
    double *sendA = nullptr, *recvA = nullptr;
    int* dMultiSlots = nullptr;          // 16 per listed point: recv slot of each sharer, -1 this rank, -2 none
    int* dPeer = nullptr;                // two-sharer points: the other rank's receive slot | selfFirst << 30 (k_halo_combineA2)
    int* dMultiIdx = nullptr;            // shared points with 3..16 sharers (combineMulti, the trailing workgroups of k_halo_combineA)
    int nMulti = 0;
    double *dOwnL = nullptr, *dCombL = nullptr, *sendL = nullptr, *recvL = nullptr;   // boundary layer treatment under -parallel
    std::vector<int> sharedLocalHost;
    std::unique_ptr<LayerBuilder> lb;                                                  // set-up in progress
    std::vector<double> lbArea;
    std::vector<uint8_t> lbInternal;
    int *sendF = nullptr, *recvF = nullptr;
    double* localStats = nullptr;
    double* statsHistory = nullptr;      // smgpu_halo_set_stats_history: ring of {residual, nFrozenPoints} records
    int32_t statsHistoryCap = 0;
    int64_t statsHistoryN = 0;
    int haloIter = 0;
    // peer-store transport (smgpu_halo_set_push): the peers' mapped receive buffers and flag words, this rank's tables
    bool pushOn = false;
    int pushPeers = 0, pushStride = 0;
    std::vector<int32_t> pushCount, pushRemoteBase, pushMyIndex;
    std::vector<void*> pushRecvA, pushRecvL, pushRecvF, pushFlags;
    void* pushLocalFlags = nullptr;
    unsigned long long pushTimeoutTicks = 6000000000ull;   // bounded wait for a peer's flag (100 MHz ticks; SMGPU_PUSH_TIMEOUT_S)
    void *dSlotA = nullptr, *dSlotL = nullptr, *dSlotF = nullptr, *dPeerFlag = nullptr;
    unsigned* dPushTicket = nullptr;
    int *dInteriorTiles = nullptr, *dSharedTiles = nullptr;   // smoothing tiles without / with shared points
    int nInteriorTiles = 0, nSharedTiles = 0;
    bool interiorDone = false;
    int *dGeomInterior = nullptr, *dGeomShared = nullptr;     // geometry tiles without / with a shared point
    int nGeomInterior = 0, nGeomShared = 0;
    bool geomAheadDone = false;
    // launches whose workgroups play several roles (k_geom_halo / k_smooth_halo, kernels_tiled.hpp): constraints off, in order
    int* dPosSlot = nullptr;                        // per smoothing-tile position: shared-point slot or -1
    SmoothTiles shr;                                // tiles over the shared points only (the halo roles run on these)
    SmoothTileView hv{};
    size_t haloLds = 0;
    std::vector<uint8_t> isInternalHost;            // findInternalMeshPoints' mask as given to smgpu_create (the tile builder marks internal neighbours)
    unsigned* dRoleTickets = nullptr;               // two sets of kRoleWords words (kernels.hpp roleDone): k_geom_halo's, k_smooth_halo's
    unsigned roleLaunches[3] = {0, 0, 0};
    int* dOwnF = nullptr;                           // local freeze flags of the shared points (k_smooth_halo's fix role)
    bool fixInSmooth = false;                       // this iteration's k_shared_fix work went out inside k_smooth_halo
    bool mergedWanted = true;                       // SMGPU_HALO_MERGED=0: the one-kernel-per-step form (the A/B)
    // the "flagged" arrangement of the multi-role launches with an exchange stream: the host's exchanges are ordered against the
    // kernels by FLAG WORDS instead of kernel boundaries -- the role that packs an exchange raises a word the exchange stream
    // waits for (hipStreamWaitValue32), the exchange stream writes a word (hipStreamWriteValue32) the consuming role polls --
    // so both exchanges run next to the bulk of the two launches.  Records leave through write-through stores into the send
    // buffers (a PushView onto this rank's own buffers).
    bool flagWanted = true;                         // SMGPU_HALO_FLAGGED=0: exchange stream ordered by events / stream ops around whole launches
    bool flagBuilt = false;
    uint32_t* dFlagWords = nullptr;                 // [0], [1]: raised by the kernels (A packed, F packed); [16], [17]: written by the exchange stream
    void *dSelfSlotA = nullptr, *dSelfSlotF = nullptr, *dSelfPeerFlag = nullptr;
    unsigned* dSelfTicket = nullptr;
    PushView flagView{};
    bool mergedIter = false;                        // this iteration's geometry + pack went out as k_geom_halo
    // LDS staging tiles (tiles.hpp)
    bool useTiles = false;
    int geomT = 128, smoothT = 256;
    GeomTiles gt;
    GeomTilesDev gtDev;        // the geometry tile tables of a device build (tiles_dev.hip): read where they were built
    EdgeTilesDev etDev;        // ... and the edge tile tables
    SmoothTilesDev stDev;      // ... and the smoothing tile tables
    DeviceTopologyArrays devLists;   // (not owning: the adopted arrays of a device build, for downloadDeferredLists)
    SmoothTiles stl;
    GeomTileView gv{};
    SmoothTileView sv{};
    size_t geomLds = 0, smoothLds = 0;
    bool writeFaces = false;   // debug: publish per-face centres/areas from the tiled geometry kernel
    // face-angle walk: compacted tables + host replay (kernels_walk.hpp) when many points are active
    int walkMode = -1;         // -1 undecided, 0 one-wave device replay (k_fa_pred + k_fa_walk: few active points), 1 host replay,
                               // 2 device replay as a causal fixed point (k_walk_fix: many active points)
    bool walkForced = false;   // SMGPU_WALK / SMGPU_HOST_WALK name the replay form: no re-decision
    int walkSwitches = 0;      // how often the automatic choice changed since the parameters were set (smgpu_debug_walk_mode)
    int* nActiveHost = nullptr;   // pinned, device-visible: nActive of the last iteration the GPU has finished (-1: none yet)
    long walkDecisions = 0;
    hipEvent_t evWalkLag[2] = {nullptr, nullptr};
    unsigned long long* dWalkOps = nullptr;   // [64] algorithmic FP64 instructions of the walk predicates, counted in timing passes
    unsigned long long* dWalkMemo = nullptr;  // SMGPU_WALK_MEMO_STATS=1: [0] matches, [1] stars, [2 + p] hash of the star inputs of point p (k_walk_pred_pack)
    unsigned long long walkMemoLast[2] = {0, 0};
    bool walkAlloc = false;
    WalkView wv{};
    FixView fxw{};             // state of the device replay (walkMode 2, k_walk_fix)
    bool fixAlloc = false;
    int walkFixBlocks = 128;
    int deviceShare = 1;          // engines computing on this device at the same time (smgpu_set_device_share)
    double* dStepSqr = nullptr;   // |proposal - current|^2 per point (k_apply_swap)
    int walkSweeps = 8;        // SMGPU_WALK_SWEEPS: sweeps of a workgroup over its slab of the item sequence between two grid barriers
    bool bndInGeom = true;     // SMGPU_BND_IN_GEOM=0: boundary pre-kernels on a side stream / in order instead of inside the geometry launch
    bool bndPreDone = false;
    bool faSideExact = true;   // SMGPU_FA_SIDE_EXACT=0: the exact face-angle pass on the main stream after the edge-angle kernels
    bool faExactOnSide = false;
    bool walkPack = true;      // SMGPU_WALK_PACK=0: k_walk_pred_star (one job per step on all ring places) instead of k_walk_pred_pack (the jobs' touched places packed over the wave)
    bool walkCache = true;     // SMGPU_WALK_CACHE=0: every star staged from the addressing in every iteration (k_walk_pred_pack) instead of from its record (StarCache)
    StarCache starCache{nullptr, nullptr, nullptr, 0};
    int starBlocks = 256 * 32;   // workgroups of k_walk_pred_star (each wave walks the active points with this stride)
    bool faLists = true;       // SMGPU_FA_LISTS=0: exact face-angle kernels over all edges / points asking the filter's marks
    bool walkStar = true;      // SMGPU_WALK_STAR=0: per-entry gather form of the walk predicates (k_walk_pred_self + k_walk_pred)    // SMGPU_WALK_BLOCKS: workgroups of the persistent replay launch (all must be resident at once)
    int walkBlocks = 0;
    void* pinned = nullptr;
    size_t pinnedBytes = 0;
    std::vector<uint8_t> walkFrozen;
    std::vector<int> walkStack, walkOut;
    std::vector<char> walkHost;
    // f32 filters in front of the two angle evaluators (kernels_filter.hpp); SMGPU_FILTER=0 disables
    bool useFilter = true, exactAll = false;
    uint8_t *dFaMaybe = nullptr, *dEaMaybe = nullptr;
    // the face-angle filter only needs the geometry, so it runs on a side stream next to the proposal kernel
    hipStream_t side = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // cross-stream ordering by stream memory operations (hipStreamWriteValue32 on the producer, hipStreamWaitValue32 on the
    // consumer, one monotonically increasing word per dependency kind): measured 4.1 us per dependency against 10.1 us for
    // hipEventRecord + hipStreamWaitEvent (scripts/native/stream_dep_bench.hip).  SMGPU_STREAM_OPS=0 selects the events.
    uint32_t* depWords = nullptr;        // one word per dependency kind, 64 bytes apart
    uint32_t depValue[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool streamOps = false;
    // multi-rank: the stream the host enqueues its exchanges on (smgpu_halo_desc.exchangeStream) and the events
    // that order it against the engine's stream
    int deferN = 0, deferIter = 0;   // close the previous iteration inside the next geometry launch (k_geom_tile)
    double* deferLocal = nullptr;    // multi-rank: where that deferred reduction leaves {residual, nFrozenPoints}
    double* deferHist = nullptr;
    bool foamOrg = false;      // OpenFOAM.org 12's face / cell geometry formulas instead of OpenFOAM.com's (smgpu_set_foam_variant)
    int xcdMap = 1;            // SMGPU_XCD_MAP: contiguous tile range per XCD (L2 sharing between neighbouring tiles)
    bool layersOn = false;     // smgpu_set_layers
    bool packTiles = true;     // SMGPU_PACK_TILES=0: exchange A packed by the per-point kernel (k_halo_packA)
    bool bndOn = false;        // smgpu_set_boundary_smoothing
    hipStream_t bndSide = nullptr;   // k_bnd_normals / k_bnd_feature run next to the geometry kernel
    hipEvent_t evBndFork = nullptr, evBndJoin = nullptr;
    bool bndPreInFlight = false;
    bool bndPending = false, bndNormalsFromLayers = false;   // step-wise set-up state
    std::vector<double> bndPts, bndSurfPts;
    std::vector<int32_t> bndSurfTris;
    std::vector<uint8_t> bndFlags, bndInternal;
    std::vector<BndPatch> bndPatches;
    double bndBlend = 0.0;
    BndView bv{};
    BoundarySetup bs;
    std::vector<int32_t> layerHopsHost, layerMapHost;   // kept for the debug getters
    bool useExch = false;
    hipStream_t exch = nullptr;
    hipEvent_t evToExch = nullptr, evFromExch = nullptr;
    bool faFilterInFlight = false;
    EdgeTiles etl;
    EdgeTileView ev{};
    bool edgeTilesOk = false;
    size_t avgPackedCount = 0;   // > 0: fAvg is stored in geometry-tile order (State::avgPacked)
    size_t edgeLds = 0;
    bool eaCoop = true;        // wave-cooperative edge-angle kernel (SMGPU_EDGE_ANGLE=faithful selects the per-angle acos form)
    int eaMaxEntries = 0;
};

static int envInt(const char* name, int def) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : def;
}

// rocprofv3 --pmc serialises kernels: a stream that waits for a value written behind a kernel of another stream would never
// wake up (observed as a hang until the outer time limit).  Under counter collection the side streams stay off by default.
// Otherwise the default follows the mesh size (round 4, measured with every kernel also timed alone, profiles/r4/side_stream_ab.txt):
// on the 1 M-cell meshes the two evaluator chains side by side win 3-4 % (hex100c 0.231 against 0.239 ms, cavity100c 0.716 against
// 0.747) -- their kernels are short, one chain's ramp-up and tail hide behind the other -- but on the 10 M-cell mesh every kernel
// fills the chip on its own, both chains are FP64-issue bound, and sharing it costs 2.7 % (3.399 against 3.308 ms per iteration).
static int sideStreamDefault(int64_t nPoints) {
    const char* v = std::getenv("ROCPROF_COUNTER_COLLECTION");
    if (v && std::atoi(v) != 0) return 0;
    return nPoints <= (int64_t)envInt("SMGPU_SIDE_STREAM_MAX_POINTS", 4000000) ? 1 : 0;
}


template <typename T>
static int devAlloc(smgpu_handle* h, T** out, size_t n) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_OK(hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    h->deviceBytes += (int64_t)bytes;
    *out = (T*)p;
    return 0;
}
template <typename T>
static int devUpload(smgpu_handle* h, const T** out, const std::vector<T>& v) {
    T* p = nullptr;
    if (devAlloc(h, &p, v.size())) return 1;
    if (!v.empty()) HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

static Prm makePrm(const smgpu_handle* h) {
    const smgpu_params& p = h->prm;
    Prm r;
    r.layersOn = h->layersOn ? 1 : 0;
    r.bndOn = h->bndOn ? 1 : 0;
    r.maxStep = p.maxStepLength;
    r.relStepFrac = p.relStepFrac;
    r.minEdge = p.minEdgeLength;
    r.totalMinFreeze = p.totalMinFreeze;
    r.smallAngle = SMGPU_PI * p.minAngle / 180.0;   // SM.C:921,1364
    r.largeAngle = SMGPU_PI * p.maxAngle / 180.0;   // SM.C:1365
    // the f32 face-angle filter tests cos(angle sum): GOOD for sure iff faCosHi < cos < faCosLo (and the sum is below pi); out-of-range
    // parameters leave it undecided (every edge then takes the exact path); NaN compares false everywhere
    const double M = (double)kFaMargin + 2.0e-5;
    r.faCosLo = (r.smallAngle >= SMGPU_PI) ? -2.0f : (float)(std::cos(std::max(r.smallAngle, 0.0)) - M);
    r.faCosHi = (r.largeAngle <= 0.0) ? 2.0f : (float)(std::cos(std::min(r.largeAngle, SMGPU_PI)) + M);
    static const long long nearUlps = [] { const char* e = std::getenv("SMGPU_NEARTIE_ULPS"); return e ? std::max(0ll, std::atoll(e)) : 4ll; }();   // (tests widen the window to see the census count)
    r.nearUlps = nearUlps;
    return r;
}

static inline int gridFor(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

// Launch helper: optional hipEvent bracketing on the handle's stream.
template <typename F>
static int launchK(smgpu_handle* h, int k, F&& f, hipStream_t stream = nullptr) {
    if (!stream) stream = h->stream;
    if (h->timing) {
        hipEvent_t a, b;
        for (hipEvent_t* ev : {&a, &b}) {
            if (!h->freeEvents.empty()) { *ev = h->freeEvents.back(); h->freeEvents.pop_back(); }
            else HIP_OK(hipEventCreate(ev));
        }
        HIP_OK(hipEventRecord(a, stream));
        f();
        HIP_OK(hipEventRecord(b, stream));
        h->pending.push_back({k, a, b});
    } else {
        f();
    }
    // a bad launch configuration is reported through the sticky last-error, not by the launch macro
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(std::string("launch of ") + kKernelNames[k] + ": " + hipGetErrorString(le));
    h->launches[k]++;
    return 0;
}

// The same for ONE kernel launched through hipExtLaunchKernelGGL: the two events are attached to the dispatch itself, so
// their difference is the kernel's own start-to-end time (what rocprofv3 reports) -- an event pair recorded around a launch
// adds the handling of two marker packets (~5 us) to it.  f(start, stop) launches; both are NULL when timing is off.
template <typename F>
static int launchKDispatch(smgpu_handle* h, int k, F&& f) {
    if (h->timing) {
        hipEvent_t a, b;
        for (hipEvent_t* ev : {&a, &b}) {
            if (!h->freeEvents.empty()) { *ev = h->freeEvents.back(); h->freeEvents.pop_back(); }
            else HIP_OK(hipEventCreate(ev));
        }
        f(a, b);
        h->pending.push_back({k, a, b});
    } else {
        f((hipEvent_t) nullptr, (hipEvent_t) nullptr);
    }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(std::string("launch of ") + kKernelNames[k] + ": " + hipGetErrorString(le));
    h->launches[k]++;
    return 0;
}

// memset on the null stream + wait: the engine's streams are non-blocking (not ordered with the null stream), and a memset of
// device memory may return before it has run -- anything a kernel on those streams reads has to be complete here
static hipError_t zeroNow(void* p, int v, size_t bytes) {
    hipError_t e = hipMemset(p, v, bytes);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

enum { DEP_FORK = 0, DEP_JOIN = 1, DEP_TO_EXCH = 2, DEP_FROM_EXCH = 3, DEP_BND_FORK = 4, DEP_BND_JOIN = 5, DEP_COUNT = 6 };
static int depInit(smgpu_handle* h) {
    if (h->depWords || !envInt("SMGPU_STREAM_OPS", 1)) return 0;
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, h->device) != hipSuccess || !can) { (void)hipGetLastError(); return 0; }
    if (hipMalloc((void**)&h->depWords, DEP_COUNT * 64) != hipSuccess) { (void)hipGetLastError(); h->depWords = nullptr; return 0; }
    if (zeroNow(h->depWords, 0, DEP_COUNT * 64) != hipSuccess) return fail("hipMemset failed");
    h->streamOps = true;
    return 0;
}
// `from` has reached this point  ==>  everything enqueued on `to` afterwards may start.
// A failing stream memory operation switches the handle to events for good: the pair (signal, wait) of one dependency
// always uses the same mechanism because the switch happens inside depSignal, before its wait is enqueued.
static int depSignal(smgpu_handle* h, int kind, hipStream_t from, hipEvent_t ev) {
    if (h->streamOps) {
        if (hipStreamWriteValue32(from, h->depWords + 16 * kind, ++h->depValue[kind], 0) == hipSuccess) return 0;
        (void)hipGetLastError();
        h->streamOps = false;
    }
    HIP_OK(hipEventRecord(ev, from));
    return 0;
}
static int depWait(smgpu_handle* h, int kind, hipStream_t to, hipEvent_t ev) {
    if (h->streamOps) { HIP_OK(hipStreamWaitValue32(to, h->depWords + 16 * kind, h->depValue[kind], hipStreamWaitValueGte, 0xffffffffu)); }
    else HIP_OK(hipStreamWaitEvent(to, ev, 0));
    return 0;
}

static int drainTimers(smgpu_handle* h) {
    if (h->pending.empty()) return 0;
    HIP_OK(hipStreamSynchronize(h->stream));
    if (h->side) HIP_OK(hipStreamSynchronize(h->side));
    if (h->bndSide) HIP_OK(hipStreamSynchronize(h->bndSide));
    for (auto& p : h->pending) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, p.a, p.b));
        h->ms[p.k] += ms;
        h->freeEvents.push_back(p.a);
        h->freeEvents.push_back(p.b);
    }
    h->pending.clear();
    return 0;
}

// the single-rank loop's restore step as a pointer swap + k_apply_swap (see smgpu_iterate)
static bool swapApplyOn(const smgpu_handle* h) {
    return (h->prm.edgeAngleConstraint || h->prm.faceAngleConstraint) && h->useTiles && !h->bndOn && !h->haloOn && h->dStepSqr &&
           envInt("SMGPU_APPLY_SWAP", 1);
}

// ... and in the step-wise multi-rank loop with the constraints on (smgpu_iter_end): the OR of the received freeze flags first
// (k_halo_orF, SM.C:2374), then the same pointer swap -- k_apply cost 117 us per iteration on configs[4]'s rank against 40
static bool swapApplyHalo(const smgpu_handle* h) {
    return (h->prm.edgeAngleConstraint || h->prm.faceAngleConstraint) && h->useTiles && !h->bndOn && !h->layersOn && h->haloOn && h->dStepSqr &&
           envInt("SMGPU_APPLY_SWAP", 1);
}

static void computeAlgoBytes(smgpu_handle* h) {
    const Topology& t = h->topo;
    const int64_t P = t.nPoints, C = t.nCells, F = t.nFaces, E = t.nEdges;
    const int64_t nfp = t.facePoints.nnz(), npc = t.pointCells.nnz(), npp = (int64_t)t.pointPoints.size(), npf = t.facePoints.nnz();      // (sizes of lists that are on the host in every build)
    const int64_t nef = t.edgeFaces.nnz(), nec = t.edgeCells.nnz(), ncf = t.cellFacesGeom.nnz();
    const bool fa = h->prm.faceAngleConstraint;
    int64_t* b = h->algoBytes;
    b[K_FACE_GEOM] = 24 * P + 4 * (F + 1) + 4 * nfp + 48 * F + (fa ? 24 * F : 0);
    b[K_CELL_CENTRES] = 4 * (C + 1) + 4 * ncf + 48 * F + 24 * C;
    const int64_t smooth = 4 * (P + 1) + 4 * npc + 24 * C + 4 * (P + 1) + 4 * npp + 24 * P + P + 24 * P;
    b[K_SMOOTH_FINAL] = smooth;
    b[K_SMOOTH_PROP] = smooth + P + (swapApplyOn(h) ? 8 * P : 0);
    b[K_EDGE_ANGLE] = 4 * (P + 1) + 8 * npf + 24 * P + 24 * P + 2 * P;
    b[K_FA_EDGES] = 8 * E + 4 * (E + 1) + 4 * nef + 4 * (E + 1) + 6 * nec + 24 * P + 24 * F + 24 * C + 16 * E;
    b[K_FA_POINTS] = 4 * (P + 1) + 4 * npp + 16 * E + 16 * P + P;
    b[K_FA_PRED] = P;   // good meshes: the flag scan only
    b[K_FA_WALK] = P;
    b[K_APPLY] = swapApplyOn(h) ? 8 * P + 2 * P : 24 * P + 24 * P + 2 * P + 24 * P;   // (+ 48 bytes per restored point, not counted)
    b[K_FINISH] = 64;
    b[K_HALO] = 0;
    b[K_BND] = 0;
    // fused geometry: points + face/cell index lists + cell centres (no face arrays round trip)
    b[K_GEOM_TILE] = 24 * P + 4 * (F + 1) + 4 * nfp + 4 * (C + 1) + 4 * ncf + 24 * C + (fa ? 24 * F : 0);
    b[K_EA_FILTER] = b[K_EDGE_ANGLE];
    b[K_FA_FILTER] = b[K_FA_EDGES] - 16 * E + E + 4 * (P + 1) + 4 * npp + P;
    // FP64 VALU instructions of the geometry (makeFaceCentresAndAreas + makeCellCentresAndVols, each face once):
    // sqrt counted as 22 and a division as 11 instructions (their IEEE expansions), a 3-vector op as 3.
    //  face, n > 3 vertices: vertex average 3(n-1) + (n power of two ? 3 : 33); per fan triangle 58 (two differences 6,
    //    cross 9, magnitude 5 + sqrt, centre sum 6, three accumulations 3 + 1 + 6); result 3 + 33 + 3
    //  triangle: 6 + 3 + 6 + 9 + 3;   cell with k faces: average 3(k-1) + (3 or 33); per face 24; result 33
    int64_t ops = 0;
    for (int32_t f = 0; f < t.nFaces; ++f) {
        const int64_t n = t.facePoints.off[f + 1] - t.facePoints.off[f];
        ops += (n == 3) ? 27 : 3 * (n - 1) + (((n & (n - 1)) == 0) ? 3 : 33) + 58 * n + 39;
    }
    for (int32_t c = 0; c < t.nCells; ++c) {
        const int64_t k = t.cellFacesGeom.off[c + 1] - t.cellFacesGeom.off[c];
        ops += 3 * (k - 1) + (((k & (k - 1)) == 0) ? 3 : 33) + 24 * k + 33;
    }
    for (int k = 0; k < K_COUNT; ++k) h->algoF64[k] = 0;
    h->algoF64[K_GEOM_TILE] = ops;
}

// the device view of a set of smoothing tiles (the shared points' own tiles, smgpu_halo_configure)
// Records a tile may stage in total: its LDS footprint, which is what sets the kernels' occupancy (the LDS is allocated in
// 512-byte steps).  Smoothing tiles of 256 points: 1 112 records of 24 bytes = 26.7 KB -- SIX workgroups per CU; the greedy pass
// otherwise reaches 512 cells + 619..623 neighbours on the large meshes = 27.2 KB and five (same box, alternating three times: gather
// kernel 335.4 -> 329.4 us on the 10 M-cell polyhedral mesh, 340.1 -> 333.0 us on hex215; the 1 M-cell block sits below by itself).
// Edge tiles of 256 edges: 852 records = 20 448 B -- EIGHT workgroups of the face-angle filter per CU instead of seven (the
// face-angle group 716 -> 705 us on the 10 M-cell mesh, the iteration 3.066 -> 3.047 ms; a tighter cap makes too many tiles: 720 us).
// largest sum over the tiles of the entries of some per-tile offset lists (a tile's staged records of all kinds)
static size_t maxTileTotal(int nTiles, std::initializer_list<const std::vector<int32_t>*> offs) {
    size_t mx = 0;
    for (int t = 0; t < nTiles; ++t) {
        size_t n = 0;
        for (const std::vector<int32_t>* o : offs) n += (size_t)((*o)[(size_t)t + 1] - (*o)[(size_t)t]);
        mx = std::max(mx, n);
    }
    return mx;
}
// Geometry tiles: 3 x points + 7 x faces doubles; 3 980 = 31 840 B -- five workgroups per CU (32 256 B each with the kernel's 48
// static bytes at the 512-byte granularity), which the kernel's 96 VGPRs (SMGPU_GEOM_WAVES = 5) allow.  A hexahedral tile of 128
// cells (~250 points, ~450 faces = 3 900) fits as it is; the polyhedral ones are cut a little earlier.
static int defaultGeomCapWeighted(int T) { return (T == 256 && SMGPU_GEOM_WAVES >= 5) ? 3980 : 0x7fffffff; }
static int defaultSmoothCapTotal(int T) { return T == 256 ? 1112 : 0x7fffffff; }
static int defaultEdgeCapTotal() { return 852; }

static int uploadSmoothView(smgpu_handle* h, const SmoothTiles& st, SmoothTileView& v, int usePairShare) {
    int rc = 0;
    rc |= devUpload(h, &v.ptOrder, st.order);
    rc |= devUpload(h, &v.ptBeg, st.ptBeg);
    rc |= devUpload(h, &v.tcOff, st.tcOff);
    rc |= devUpload(h, &v.tcIds, st.tcIds);
    rc |= devUpload(h, &v.tnOff, st.tnOff);
    rc |= devUpload(h, &v.tnIds, st.tnIds);
    rc |= devUpload(h, &v.selfLoc, st.selfLoc);
    rc |= devUpload(h, &v.pcBase, st.pcBase);
    rc |= devUpload(h, &v.pcWidth, st.pcWidth);
    rc |= devUpload(h, &v.pcEll, st.pcEll);
    rc |= devUpload(h, &v.ppBase, st.ppBase);
    rc |= devUpload(h, &v.ppWidth, st.ppWidth);
    rc |= devUpload(h, &v.ppEll, st.ppEll);
    rc |= devUpload(h, &v.pairEll, st.pairEll);
    rc |= devUpload(h, &v.pfBase, st.pfBase);
    rc |= devUpload(h, &v.pfWidth, st.pfWidth);
    rc |= devUpload(h, &v.pfEll, st.pfEll);
    std::vector<int> meta((size_t)kSmoothMetaInts * (size_t)std::max(st.nTiles, 1), 0);
    for (int t = 0; t < st.nTiles; ++t) {
        int* r = meta.data() + (size_t)kSmoothMetaInts * t;
        r[0] = st.ptBeg[t]; r[1] = st.ptBeg[t + 1] - st.ptBeg[t]; r[2] = st.tcOff[t]; r[3] = st.tcOff[t + 1] - st.tcOff[t];
        r[4] = st.tnOff[t]; r[5] = st.tnOff[t + 1] - st.tnOff[t]; r[6] = st.pcBase[t]; r[7] = st.pcWidth[t];
        r[8] = st.ppBase[t]; r[9] = st.ppWidth[t]; r[10] = st.pfBase[t]; r[11] = st.pfWidth[t];
    }
    rc |= devUpload(h, &v.meta, meta);
    v.maxCells = st.maxCells; v.maxPoints = st.maxPoints;
    v.usePairShare = usePairShare;
    return rc;
}

extern "C" {

const char* smgpu_last_error(void) { return g_err.c_str(); }
const char* smgpu_version(void) { return "smgpu 0.1 (gfx950)"; }

// the lists of a device build that stayed on the device only (topology.hpp, downloadDeferredLists), for the few host readers
static int ensureHostLists(smgpu_handle* h, int groups) {
    if (!h->devLists.valid) return 0;
    std::string why;
    return downloadDeferredLists(h->topo, h->devLists, h->device, why, groups) ? fail("device -> host copy of the addressing: " + why) : 0;
}

// A device build's owner / neighbour arrays are read by the device builds of the tile tables only and never join the handle's
// allocation list: freed once those builds are joined (and on every failure path).
static void freeOwnerNeighbour(DeviceTopologyArrays& dt) {
    for (DeviceTopologyArrays::Arr* a : {&dt.owner, &dt.neighbour})
        if (a->p) { (void)hipFree(a->p); a->p = nullptr; a->bytes = 0; }
}
static void freeDevTopo(DeviceTopologyArrays& dt) {
    freeOwnerNeighbour(dt);
    for (DeviceTopologyArrays::Arr* a : {&dt.faceOff, &dt.facePts, &dt.cfOff, &dt.cfVal, &dt.pcOff, &dt.pcVal, &dt.ppOff, &dt.ppPt, &dt.peEdge, &dt.pfOff, &dt.pfFace,
                                         &dt.pfPrev, &dt.pfNext, &dt.pfPrevSlot, &dt.pfNextSlot, &dt.ringFace, &dt.ringCell, &dt.edgeRingOk, &dt.edges, &dt.efOff,
                                         &dt.efFace, &dt.ecOff, &dt.ecCell, &dt.ecF0, &dt.ecF1})
        if (a->p) { (void)hipFree(a->p); a->p = nullptr; a->bytes = 0; }
    dt.valid = false;
}

int smgpu_create(const smgpu_mesh_desc* d, smgpu_handle** out) {
    if (!d || !out) return fail("smgpu_create: null argument");
    *out = nullptr;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return fail("smgpu_create: no HIP device available");
    if (d->device < 0 || d->device >= nDev) return fail("smgpu_create: device ordinal out of range");
    smgpu_handle* h = new smgpu_handle();
    h->device = d->device;
    const auto tCreate0 = std::chrono::steady_clock::now();
    setupClockStart() = tCreate0;
    auto sinceCreate = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tCreate0).count(); };
    // The one-off host work is pipelined: the Z-curve of the points (needs the coordinates only) starts at once; the geometry
    // tile tables start as soon as the cell -> face lists stand (Topology::build's afterCells hook) and are built next to the
    // rest of the addressing; the smoothing and edge tile tables follow the addressing, side by side.
    const bool wantTiles = envInt("SMGPU_TILES", 1) != 0;
    const bool mortonTiles = envInt("SMGPU_TILE_MORTON", 1) != 0;
    std::future<std::vector<int32_t>> fPointOrder;
    if (wantTiles && mortonTiles) fPointOrder = std::async(std::launch::async, [&] { return mortonOrderOf(d->nPoints, d->points); });
    std::future<std::string> fGeom, fSmooth, fEdge;
    std::function<void()> startEdge;
    CornerChains chains;       // (tiles_dev.hip; released by the smoothing task's device build, or below)
    struct ChainsGuard { CornerChains& c; std::future<std::string>& f; ~ChainsGuard() { if (f.valid()) f.wait(); releaseCornerChains(c); } } chainsGuard{chains, fSmooth};
    static const char* const kHostTablesPending = "\x01host tables pending";      // a tile task of a device build that leaves its tables to the host
    std::vector<int32_t> pointOrder;
    std::vector<uint8_t> internalMask;
    DeviceTopologyArrays devTopo;      // a device build's arrays: the kernels read them as they are (no upload of the host copy)
    {
        const int geomT0 = envInt("SMGPU_GEOM_T", 256), smoothT0 = envInt("SMGPU_SMOOTH_T", 256);
        const int geomCells0 = envInt("SMGPU_GEOM_CELLS", geomT0 / 2);
        const int capGP0 = envInt("SMGPU_GEOM_CAPP", std::min(6 * geomCells0, 1400)), capGF0 = envInt("SMGPU_GEOM_CAPF", std::min(4 * geomCells0, 1400));
        const int capSC0 = envInt("SMGPU_SMOOTH_CAPC", std::min(2 * smoothT0, 1500)), capSN0 = envInt("SMGPU_SMOOTH_CAPN", std::min(3 * smoothT0, 1500));
        const bool geomOk = geomT0 == 64 || geomT0 == 128 || geomT0 == 256, smoothOk = smoothT0 == 64 || smoothT0 == 128 || smoothT0 == 256;
        // tile boundaries on the host, the tables on the device where the addressing was built there (tiles_dev.hip), else on the host
        // (SMGPU_DEVICE_TILES=2: as if the device builds handed their tables back -- the tests' way into that path)
        const bool devTiles = envInt("SMGPU_DEVICE_TILES", 1) != 0;
        const auto afterCells = [&] {
                      // (the corner chains of the smoothing tables need the device addressing only: started here, they run beside the
                      // rest of the download and the host's boundary passes)
                      if (wantTiles && geomOk && smoothOk && devTiles && devTopo.valid && envInt("SMGPU_DEVICE_TILES", 1) == 1) {
                          std::string why;
                          if (startCornerChains(devTopo, d->nPoints, h->device, chains, why) == 2) chains = CornerChains();
                      }
                      if (wantTiles && geomOk) fGeom = std::async(std::launch::async, [&, geomT0, geomCells0, capGP0, capGF0, devTiles]() -> std::string {
                      std::vector<int32_t> cellOrder;      // (a device build: the Z-curve of the cells there too -- the order the host would find)
                      if (mortonTiles && devTiles && devTopo.valid && envInt("SMGPU_DEVICE_TILES", 1) == 1) {
                          std::string why;
                          if (cellMortonOrderOnDevice(devTopo, d->nCells, d->nPoints, d->points, h->device, cellOrder, why) != 0) cellOrder.clear();
                      }
                      const std::string e = h->gt.buildBoundaries(h->topo, d->points, mortonTiles, geomT0, geomCells0, capGP0, capGF0,
                                                                  envInt("SMGPU_GEOM_CAPWEIGHTED", defaultGeomCapWeighted(geomT0)), SMGPU_GEOM_AOS ? kGF : 6,
                                                                  cellOrder.empty() ? nullptr : &cellOrder);
                      if (!e.empty()) return e;
                      if (devTopo.valid && !devTiles) return std::string(kHostTablesPending);      // (the host's lists are still arriving)
                      if (devTopo.valid) {
                          std::string why;
                          const auto t0 = std::chrono::steady_clock::now();
                          const int rc = envInt("SMGPU_DEVICE_TILES", 1) == 2 ? 1 : buildGeomTablesOnDevice(h->gt, devTopo, d->nCells, h->device, h->gtDev, why);
                          if (envInt("SMGPU_VERBOSE", 0) >= 2)
                              std::fprintf(stderr, "[smgpu] geometry tiles: tables on the %s %.2f s   (done at +%.3f s)\n", rc == 0 ? "device" : "host (device build handed them back)",
                                           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), setupClock());
                          if (rc == 0) return std::string();
                          return rc == 2 ? "device tile tables: " + why : std::string(kHostTablesPending);      // (the host's lists are still arriving)
                      }
                      return h->gt.buildTables(h->topo); }); };
        const auto afterPoints = [&] {
                if (!(wantTiles && geomOk && smoothOk)) return;
                if (fPointOrder.valid()) pointOrder = fPointOrder.get();
                internalMask.resize((size_t)d->nPoints);
                for (int p = 0; p < d->nPoints; ++p) internalMask[(size_t)p] = d->isInternalPoint[p] ? 1 : 0;
                h->isInternalHost = internalMask;
                fSmooth = std::async(std::launch::async, [&, smoothT0, capSC0, capSN0, devTiles]() -> std::string {
                    const std::string e = h->stl.buildBoundaries(h->topo, d->points, mortonTiles, smoothT0, capSC0, capSN0, (mortonTiles && !pointOrder.empty()) ? &pointOrder : nullptr, nullptr,
                                                                  envInt("SMGPU_SMOOTH_CAPTOTAL", defaultSmoothCapTotal(smoothT0)));
                    if (!e.empty()) return e;
                    if (devTopo.valid && !devTiles) return std::string(kHostTablesPending);
                    if (devTopo.valid) {
                        std::string why;
                        const auto t0 = std::chrono::steady_clock::now();
                        const int rc = envInt("SMGPU_DEVICE_TILES", 1) == 2 ? 1 : buildSmoothTablesOnDevice(h->stl, devTopo, d->nPoints, h->topo.maxPointPoints, internalMask.data(), h->device, h->stDev, why, &chains);
                        if (envInt("SMGPU_VERBOSE", 0) >= 2)
                            std::fprintf(stderr, "[smgpu] smoothing tiles: tables on the %s %.2f s   (done at +%.3f s)\n", rc == 0 ? "device" : "host (device build handed them back)",
                                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), setupClock());
                        if (rc == 0) return std::string();
                        return rc == 2 ? "device tile tables: " + why : std::string(kHostTablesPending);
                    }
                    return h->stl.buildTables(h->topo, internalMask.data()); });
            };
        // the edge tiles (face-angle filter): behind the edge lists of a device build (its afterEdges hook), else behind the addressing
        startEdge = [&, devTiles] {
            if (!wantTiles) return;
            if (fPointOrder.valid()) pointOrder = fPointOrder.get();
            const std::vector<int32_t>* po = (mortonTiles && !pointOrder.empty()) ? &pointOrder : nullptr;
            const bool wantFilter = envInt("SMGPU_FILTER", 1) != 0;
            const bool devTilesE = devTiles && devTopo.valid, devLists = devTopo.valid;
            const DeviceTopologyArrays* dt = &devTopo;
            fEdge = std::async(std::launch::async, [h, d, po, wantFilter, mortonTiles, devTilesE, devLists, dt]() -> std::string {
                if (!wantFilter) return std::string("not built");
                const std::string e = h->etl.buildBoundaries(h->topo, d->points, mortonTiles, 256, envInt("SMGPU_EDGE_CAPP", 512), envInt("SMGPU_EDGE_CAPF", 768), envInt("SMGPU_EDGE_CAPC", 512), po,
                                                                  envInt("SMGPU_EDGE_CAPTOTAL", defaultEdgeCapTotal()));
                if (!e.empty()) return e;
                if (devLists && !devTilesE) return std::string(kHostTablesPending);
                if (devTilesE) {
                    std::string why;
                    const auto t0 = std::chrono::steady_clock::now();
                    const int rc = envInt("SMGPU_DEVICE_TILES", 1) == 2 ? 1 : buildEdgeTablesOnDevice(h->etl, *dt, h->topo.nEdges, h->device, h->etDev, why);
                    if (envInt("SMGPU_VERBOSE", 0) >= 2)
                        std::fprintf(stderr, "[smgpu] edge tiles: tables on the %s %.2f s   (done at +%.3f s)\n", rc == 0 ? "device" : "host (device build handed them back)",
                                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), setupClock());
                    if (rc == 0) return std::string();
                    return rc == 2 ? "device tile tables: " + why : std::string(kHostTablesPending);
                }
                return h->etl.buildTables(h->topo);
            });
        };
        // the addressing on the device (topology_dev.hip: sorts + per-edge kernels, ~0.2 s for 10 M cells against 3.5 s of host
        // loops); the host build where the device path hands the mesh back (unusual meshes; it also words the reference's errors)
        std::string terr;
        int dev = envInt("SMGPU_DEVICE_TOPOLOGY", 1) ? 1 : -1;
        if (dev > 0) {
            std::string why;
            dev = buildTopologyOnDevice(h->topo, d->nPoints, d->nCells, d->nFaces, d->nInternalFaces, d->faceOffsets, d->facePoints, d->owner, d->neighbour, h->device, why,
                                        afterCells, afterPoints, &devTopo, startEdge, devTiles && wantTiles);
            if (dev == 2) {
                if (fGeom.valid()) fGeom.wait();
                if (fSmooth.valid()) fSmooth.wait();
                if (fEdge.valid()) fEdge.wait();
                if (fPointOrder.valid()) fPointOrder.wait();
                freeDevTopo(devTopo);       // (arrays the failed device build had already handed over)
                delete h;
                return fail("smgpu_create: device addressing: " + why);
            }
            if (envInt("SMGPU_VERBOSE", 0) >= 1) std::fprintf(stderr, "[smgpu] addressing on the %s (%.2f s since create)\n", dev == 0 ? "device" : "host (device path handed the mesh back)", sinceCreate());
        }
        if (dev != 0) {
            h->topo = Topology();
            terr = h->topo.build(d->nPoints, d->nCells, d->nFaces, d->nInternalFaces, d->faceOffsets, d->facePoints, d->owner, d->neighbour, afterCells, afterPoints);
        }
        if (!terr.empty()) {
            if (fGeom.valid()) fGeom.wait();
            if (fSmooth.valid()) fSmooth.wait();
            if (fPointOrder.valid()) fPointOrder.wait();
            delete h;
            return fail("smgpu_create: " + terr);
        }
        if (!fEdge.valid()) startEdge();
    }
    bool devAdopted = false;
    auto cleanup = [&](int rc) {   // (the host threads still read the handle's tables)
        if (fGeom.valid()) fGeom.wait();
        if (fSmooth.valid()) fSmooth.wait();
        if (fPointOrder.valid()) fPointOrder.wait();
        if (fEdge.valid()) fEdge.wait();
        if (!devAdopted) freeDevTopo(devTopo);      // a device build's arrays the handle has not taken over yet
        else freeOwnerNeighbour(devTopo);           // (never adopted: only the tile-table builds read them)
        smgpu_destroy(h);
        return rc;
    };
    if (hipSetDevice(h->device) != hipSuccess) return cleanup(fail("hipSetDevice failed"));
    if (d->useCallerStream) h->stream = (hipStream_t)d->stream;
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(fail("hipStreamCreate failed"));
        h->ownStream = true;
    }
    const Topology& t = h->topo;
    MeshView& m = h->mv;
    const double tTopo = sinceCreate();
    double tLap = tTopo;
    auto lapC = [&](const char* what) { if (envInt("SMGPU_VERBOSE", 0) >= 2) { const double now = sinceCreate(); std::fprintf(stderr, "[smgpu] create: %-28s %.3f s   (at +%.3f s)\n", what, now - tLap, now); tLap = now; } };
    auto cleanupE = [&](int rc0) { if (fEdge.valid()) fEdge.wait(); return cleanup(rc0); };
    m.nPoints = t.nPoints; m.nCells = t.nCells; m.nFaces = t.nFaces; m.nInternalFaces = t.nInternalFaces; m.nEdges = t.nEdges;
    std::vector<uint8_t> flags(t.nPoints);
    for (int p = 0; p < t.nPoints; ++p)
        flags[p] = (d->isInternalPoint[p] ? PF_INTERNAL : 0) |
                   ((d->isSmoothingSurfacePoint && d->isSmoothingSurfacePoint[p]) ? PF_SMOOTHSURF : 0);
    int rc = 0;
    // the addressing on the device: a device build's own arrays (topology_dev.hip), else uploads of the host build's lists
    auto adopt = [&](auto*& dst, const DeviceTopologyArrays::Arr& a) { dst = (std::remove_reference_t<decltype(dst)>)a.p; h->allocs.push_back(a.p); h->deviceBytes += (int64_t)a.bytes; };
    if (devTopo.valid) {
        devAdopted = true;
        h->devLists = devTopo;
        adopt(m.faceOff, devTopo.faceOff);
        adopt(m.facePts, devTopo.facePts);
        adopt(m.cfOff, devTopo.cfOff);
        adopt(m.cfVal, devTopo.cfVal);
        adopt(m.pcOff, devTopo.pcOff);
        adopt(m.pcVal, devTopo.pcVal);
        adopt(m.ppOff, devTopo.ppOff);
        adopt(m.ppPt, devTopo.ppPt);
        adopt(m.peEdge, devTopo.peEdge);
        adopt(m.pfOff, devTopo.pfOff);
        adopt(m.pfFace, devTopo.pfFace);
        adopt(m.pfPrev, devTopo.pfPrev);
        adopt(m.pfNext, devTopo.pfNext);
        adopt(m.pfPrevSlot, devTopo.pfPrevSlot);
        adopt(m.pfNextSlot, devTopo.pfNextSlot);
        adopt(m.ringFace, devTopo.ringFace);
        adopt(m.ringCell, devTopo.ringCell);
        adopt(m.edgeRingOk, devTopo.edgeRingOk);
        adopt(m.edges, devTopo.edges);
        adopt(m.efOff, devTopo.efOff);
        adopt(m.efFace, devTopo.efFace);
        adopt(m.ecOff, devTopo.ecOff);
        adopt(m.ecCell, devTopo.ecCell);
        adopt(m.ecF0, devTopo.ecF0);
        adopt(m.ecF1, devTopo.ecF1);
    } else {
        rc |= devUpload(h, &m.faceOff, t.facePoints.off);
        rc |= devUpload(h, &m.facePts, t.facePoints.val);
        rc |= devUpload(h, &m.cfOff, t.cellFacesGeom.off);
        rc |= devUpload(h, &m.cfVal, t.cellFacesGeom.val);
        rc |= devUpload(h, &m.pcOff, t.pointCells.off);
        rc |= devUpload(h, &m.pcVal, t.pointCells.val);
        rc |= devUpload(h, &m.ppOff, t.pointEdges.off);
        rc |= devUpload(h, &m.ppPt, t.pointPoints);
        rc |= devUpload(h, &m.peEdge, t.pointEdges.val);
        rc |= devUpload(h, &m.pfOff, t.pointFaces.off);
        rc |= devUpload(h, &m.pfFace, t.pointFaces.val);
        rc |= devUpload(h, &m.pfPrev, t.pfPrev);
        rc |= devUpload(h, &m.pfNext, t.pfNext);
        rc |= devUpload(h, &m.pfPrevSlot, t.pfPrevSlot);
        rc |= devUpload(h, &m.pfNextSlot, t.pfNextSlot);
        rc |= devUpload(h, &m.ringFace, t.ringFace);
        rc |= devUpload(h, &m.ringCell, t.ringCell);
        rc |= devUpload(h, &m.edgeRingOk, t.edgeRingOk);
        rc |= devUpload(h, &m.edges, t.edges);
        rc |= devUpload(h, &m.efOff, t.edgeFaces.off);
        rc |= devUpload(h, &m.efFace, t.edgeFaces.val);
        rc |= devUpload(h, &m.ecOff, t.edgeCells.off);
        rc |= devUpload(h, &m.ecCell, t.edgeCells.val);
        rc |= devUpload(h, &m.ecF0, t.ecFace0);
        rc |= devUpload(h, &m.ecF1, t.ecFace1);
    }
    rc |= devUpload(h, &m.pflags, flags);
    if (rc) return cleanupE(1);
    lapC("addressing adopted / uploaded");
    // LDS staging tiles; SMGPU_TILES=0 keeps the direct-gather kernels (A/B and fallback)
    h->useTiles = envInt("SMGPU_TILES", 1) != 0;
    h->useFilter = envInt("SMGPU_FILTER", 1) != 0;
    h->xcdMap = envInt("SMGPU_XCD_MAP", 1) != 0;
    h->walkStar = envInt("SMGPU_WALK_STAR", 1) != 0;
    h->starBlocks = std::max(1, envInt("SMGPU_STAR_BLOCKS", 256 * 32));
    h->walkPack = envInt("SMGPU_WALK_PACK", 1) != 0;
    h->walkCache = envInt("SMGPU_WALK_CACHE", 1) != 0;
    h->faLists = envInt("SMGPU_FA_LISTS", 1) != 0;
    h->faSideExact = envInt("SMGPU_FA_SIDE_EXACT", 1) != 0;
    h->bndInGeom = envInt("SMGPU_BND_IN_GEOM", 1) != 0;
    { const char* fv = std::getenv("SMGPU_FOAM_VARIANT"); h->foamOrg = fv && std::string(fv) == "org"; }
    { const char* sv = std::getenv("SMGPU_SYNC_VARIANT"); h->st.ownFold = (sv && std::string(sv) == "own") ? 1 : 0; }
    if (h->useTiles) {
        h->geomT = envInt("SMGPU_GEOM_T", 256);
        h->smoothT = envInt("SMGPU_SMOOTH_T", 256);
        if (h->geomT != 64 && h->geomT != 128 && h->geomT != 256) return cleanupE(fail("SMGPU_GEOM_T must be 64, 128 or 256"));
        if (h->smoothT != 64 && h->smoothT != 128 && h->smoothT != 256) return cleanupE(fail("SMGPU_SMOOTH_T must be 64, 128 or 256"));
        // capacities sized for a 160 KiB LDS: a tile must leave room for >= 2 workgroups per CU
        const int geomCells = envInt("SMGPU_GEOM_CELLS", h->geomT / 2);   // cells per tile (<= threads)
        const int capGP = envInt("SMGPU_GEOM_CAPP", std::min(6 * geomCells, 1400));
        const int capGF = envInt("SMGPU_GEOM_CAPF", std::min(4 * geomCells, 1400));   // two rounds of 256 face threads at 128 cells
        const int capSC = envInt("SMGPU_SMOOTH_CAPC", std::min(2 * h->smoothT, 1500));
        const int capSN = envInt("SMGPU_SMOOTH_CAPN", std::min(3 * h->smoothT, 1500));
        // (the geometry and smoothing tables were started by the addressing's hooks, the edge tables behind it; SMGPU_GEOM_T etc. were read there)
        (void)geomCells; (void)capGP; (void)capGF; (void)capSC; (void)capSN;
        std::string e2 = fSmooth.valid() ? fSmooth.get() : std::string("not built");
        std::string e1 = fGeom.valid() ? fGeom.get() : std::string("not built");
        std::string e3 = fEdge.valid() ? fEdge.get() : std::string("not built");
        // (a device build that handed a table set back: the host builds it now that all of its lists have arrived)
        if (e1 == kHostTablesPending) e1 = h->gt.buildTables(h->topo);
        std::string whyD;
        if (e2 == kHostTablesPending) e2 = downloadDeferredLists(h->topo, devTopo, h->device, whyD, 1) ? "device -> host copy of the addressing: " + whyD : h->stl.buildTables(h->topo, internalMask.data());
        if (e3 == kHostTablesPending) e3 = downloadDeferredLists(h->topo, devTopo, h->device, whyD, 2) ? "device -> host copy of the addressing: " + whyD : h->etl.buildTables(h->topo);
        if (envInt("SMGPU_VERBOSE", 0))
            std::fprintf(stderr, "[smgpu] set-up: addressing %.2f s, tile tables (3 host threads) %.2f s\n", tTopo, sinceCreate() - tTopo);
        freeOwnerNeighbour(devTopo);      // read by the device builds of the tile tables only, all joined now (4 (nFaces + nInternalFaces) bytes)
        lapC("tile tasks joined");
        if (!e1.empty() || !e2.empty()) {
            h->useTiles = false;   // meshes with huge cells / valences: direct-gather kernels still apply
        } else {
            GeomTileView& g = h->gv;
            SmoothTileView& v = h->sv;
            auto adoptT = [&](auto*& dst, const GeomTilesDev::Arr& a) { dst = (std::remove_reference_t<decltype(dst)>)a.p; h->allocs.push_back(a.p); h->deviceBytes += (int64_t)a.bytes; };
            if (h->gtDev.valid) {      // a device build's tables stay where they are
                adoptT(g.cellOrder, h->gtDev.cellOrder); adoptT(g.cellBeg, h->gtDev.cellBeg); adoptT(g.tpIds, h->gtDev.tpIds); adoptT(g.tfIds, h->gtDev.tfIds);
                adoptT(g.faceVerts, h->gtDev.faceVerts); adoptT(g.cellFaces, h->gtDev.cellFaces); adoptT(g.meta, h->gtDev.meta);
                h->gtDev.valid = false;      // (the handle's allocation list owns them now)
            } else {
                rc |= devUpload(h, &g.cellOrder, h->gt.order);
                rc |= devUpload(h, &g.cellBeg, h->gt.cellBeg);
                rc |= devUpload(h, &g.tpIds, h->gt.tpIds);
                rc |= devUpload(h, &g.tfIds, h->gt.tfIds);
                rc |= devUpload(h, &g.faceVerts, h->gt.faceVerts);
                rc |= devUpload(h, &g.cellFaces, h->gt.cellFaces);
            }
            rc |= devUpload(h, &g.tpOff, h->gt.tpOff);
            rc |= devUpload(h, &g.tfOff, h->gt.tfOff);
            rc |= devUpload(h, &g.fvBase, h->gt.fvBase);
            rc |= devUpload(h, &g.fvWidth, h->gt.fvWidth);
            rc |= devUpload(h, &g.cfBase, h->gt.cfBase);
            rc |= devUpload(h, &g.cfWidth, h->gt.cfWidth);
            rc |= devUpload(h, &g.tileFlags, h->gt.tileFlags);
            if (!g.meta) {   // the per-tile scalars once more, one record per tile (GeomTileMeta: scalar loads in the kernel)
                const auto& gt = h->gt;
                std::vector<int> meta((size_t)kGeomMetaInts * (size_t)gt.nTiles, 0);
                for (int t = 0; t < gt.nTiles; ++t) {
                    int* r = meta.data() + (size_t)kGeomMetaInts * t;
                    r[0] = gt.tpOff[t]; r[1] = gt.tpOff[t + 1] - gt.tpOff[t]; r[2] = gt.tfOff[t]; r[3] = gt.tfOff[t + 1] - gt.tfOff[t];
                    r[4] = gt.fvBase[t]; r[5] = gt.fvWidth[t]; r[6] = gt.cellBeg[t]; r[7] = gt.cellBeg[t + 1] - gt.cellBeg[t];
                    r[8] = gt.cfBase[t]; r[9] = gt.cfWidth[t]; r[10] = gt.tileFlags[t];
                }
                rc |= devUpload(h, &g.meta, meta);
            }
            g.maxPoints = h->gt.maxPoints; g.maxFaces = h->gt.maxFaces;
            if (h->stDev.valid) {
                auto adoptS = [&](auto*& dst, const SmoothTilesDev::Arr& a) { dst = (std::remove_reference_t<decltype(dst)>)a.p; h->allocs.push_back(a.p); h->deviceBytes += (int64_t)a.bytes; };
                adoptS(v.ptOrder, h->stDev.order); adoptS(v.ptBeg, h->stDev.ptBeg); adoptS(v.tcIds, h->stDev.tcIds); adoptS(v.tnIds, h->stDev.tnIds); adoptS(v.selfLoc, h->stDev.selfLoc);
                adoptS(v.pcEll, h->stDev.pcEll); adoptS(v.ppEll, h->stDev.ppEll); adoptS(v.pairEll, h->stDev.pairEll); adoptS(v.pfEll, h->stDev.pfEll); adoptS(v.meta, h->stDev.meta);
                h->stDev.valid = false;
            } else {
                rc |= devUpload(h, &v.ptOrder, h->stl.order);
                rc |= devUpload(h, &v.ptBeg, h->stl.ptBeg);
                rc |= devUpload(h, &v.tcIds, h->stl.tcIds);
                rc |= devUpload(h, &v.tnIds, h->stl.tnIds);
                rc |= devUpload(h, &v.selfLoc, h->stl.selfLoc);
                rc |= devUpload(h, &v.pcEll, h->stl.pcEll);
                rc |= devUpload(h, &v.ppEll, h->stl.ppEll);
                rc |= devUpload(h, &v.pairEll, h->stl.pairEll);
                rc |= devUpload(h, &v.pfEll, h->stl.pfEll);
            }
            rc |= devUpload(h, &v.tcOff, h->stl.tcOff);
            rc |= devUpload(h, &v.tnOff, h->stl.tnOff);
            rc |= devUpload(h, &v.pcBase, h->stl.pcBase);
            rc |= devUpload(h, &v.pcWidth, h->stl.pcWidth);
            rc |= devUpload(h, &v.ppBase, h->stl.ppBase);
            rc |= devUpload(h, &v.ppWidth, h->stl.ppWidth);
            rc |= devUpload(h, &v.pfBase, h->stl.pfBase);
            rc |= devUpload(h, &v.pfWidth, h->stl.pfWidth);
            if (!v.meta) {   // SmoothTileMeta records
                const auto& st = h->stl;
                std::vector<int> meta((size_t)kSmoothMetaInts * (size_t)st.nTiles, 0);
                for (int t = 0; t < st.nTiles; ++t) {
                    int* r = meta.data() + (size_t)kSmoothMetaInts * t;
                    r[0] = st.ptBeg[t]; r[1] = st.ptBeg[t + 1] - st.ptBeg[t]; r[2] = st.tcOff[t]; r[3] = st.tcOff[t + 1] - st.tcOff[t];
                    r[4] = st.tnOff[t]; r[5] = st.tnOff[t + 1] - st.tnOff[t]; r[6] = st.pcBase[t]; r[7] = st.pcWidth[t];
                    r[8] = st.ppBase[t]; r[9] = st.ppWidth[t]; r[10] = st.pfBase[t]; r[11] = st.pfWidth[t];
                }
                rc |= devUpload(h, &v.meta, meta);
            }
            v.maxCells = h->stl.maxCells; v.maxPoints = h->stl.maxPoints;
            v.usePairShare = t.maxPointPoints <= 16 ? 1 : 0;
            if (h->useFilter) {
                if (e3.empty()) {
                    EdgeTileView& ev = h->ev;
                    // the filter reads the face averages where the geometry tiles store them: position of every face
                    // in the face list of its owner's tile
                    const size_t nGeomTf = (size_t)h->gt.tfOff.back();
                    h->avgPackedCount = nGeomTf;
                    if (h->etDev.valid) {      // a device build: the remap there too, the tables stay where they are
                        std::string why;
                        if (remapEdgeFaceIdsOnDevice(g.tfIds, (long long)nGeomTf, t.nFaces, (int*)h->etDev.tfIds.p, h->etDev.nTf, h->device, why))
                            return cleanup(fail("smgpu_create: edge tile face positions: " + why));
                        auto adoptE = [&](auto*& dst, const EdgeTilesDev::Arr& a) { dst = (std::remove_reference_t<decltype(dst)>)a.p; h->allocs.push_back(a.p); h->deviceBytes += (int64_t)a.bytes; };
                        adoptE(ev.order, h->etDev.order); adoptE(ev.edgeBeg, h->etDev.edgeBeg); adoptE(ev.tpIds, h->etDev.tpIds); adoptE(ev.tfIds, h->etDev.tfIds);
                        adoptE(ev.tcIds, h->etDev.tcIds); adoptE(ev.epLoc, h->etDev.epLoc); adoptE(ev.efEll, h->etDev.efEll); adoptE(ev.ecEll, h->etDev.ecEll);
                        adoptE(ev.meta, h->etDev.meta);
                        h->etDev.valid = false;
                    } else {
                        if (h->gt.tfIds.size() != nGeomTf) {      // (a device build of the geometry tables that kept its face ids there)
                            h->gt.tfIds.resize(nGeomTf);
                            if (hipMemcpy(h->gt.tfIds.data(), g.tfIds, nGeomTf * 4, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail("smgpu_create: download of the tile face ids failed"));
                        }
                        std::vector<int32_t> facePos((size_t)t.nFaces, -1);
                        for (size_t k = 0; k < h->gt.tfIds.size(); ++k)
                            if (h->gt.tfIds[k] < 0) facePos[(size_t)(h->gt.tfIds[k] & 0x7fffffff)] = (int32_t)k;
                        for (int32_t& f : h->etl.tfIds) f = facePos[(size_t)f];
                        rc |= devUpload(h, &ev.order, h->etl.order);
                        rc |= devUpload(h, &ev.edgeBeg, h->etl.edgeBeg);
                        rc |= devUpload(h, &ev.tpIds, h->etl.tpIds);
                        rc |= devUpload(h, &ev.tfIds, h->etl.tfIds);
                        rc |= devUpload(h, &ev.tcIds, h->etl.tcIds);
                        rc |= devUpload(h, &ev.epLoc, h->etl.epLoc);
                        rc |= devUpload(h, &ev.efEll, h->etl.efEll);
                        rc |= devUpload(h, &ev.ecEll, h->etl.ecEll);
                    }
                    rc |= devUpload(h, &ev.tpOff, h->etl.tpOff);
                    rc |= devUpload(h, &ev.tfOff, h->etl.tfOff);
                    rc |= devUpload(h, &ev.tcOff, h->etl.tcOff);
                    rc |= devUpload(h, &ev.efBase, h->etl.efBase);
                    rc |= devUpload(h, &ev.ecBase, h->etl.ecBase);
                    rc |= devUpload(h, &ev.efWidth, h->etl.efWidth);
                    rc |= devUpload(h, &ev.ecWidth, h->etl.ecWidth);
                    if (!ev.meta) {   // EdgeTileMeta records
                        const auto& et = h->etl;
                        std::vector<int> meta((size_t)kEdgeMetaInts * (size_t)et.nTiles, 0);
                        for (int ti = 0; ti < et.nTiles; ++ti) {
                            int* r = meta.data() + (size_t)kEdgeMetaInts * ti;
                            r[0] = et.edgeBeg[ti]; r[1] = et.edgeBeg[ti + 1] - et.edgeBeg[ti]; r[2] = et.tpOff[ti]; r[3] = et.tpOff[ti + 1] - et.tpOff[ti];
                            r[4] = et.tfOff[ti]; r[5] = et.tfOff[ti + 1] - et.tfOff[ti]; r[6] = et.tcOff[ti]; r[7] = et.tcOff[ti + 1] - et.tcOff[ti];
                            r[8] = et.efBase[ti]; r[9] = et.efWidth[ti]; r[10] = et.ecBase[ti]; r[11] = et.ecWidth[ti];
                        }
                        rc |= devUpload(h, &ev.meta, meta);
                    }
                    ev.maxPoints = h->etl.maxPoints; ev.maxFaces = h->etl.maxFaces; ev.maxCells = h->etl.maxCells;
                    h->edgeLds = sizeof(double) * 3 * maxTileTotal(h->etl.nTiles, {&h->etl.tpOff, &h->etl.tfOff, &h->etl.tcOff});
                    h->edgeTilesOk = h->edgeLds <= 64 * 1024;
                    if (envInt("SMGPU_VERBOSE", 0))
                        std::fprintf(stderr, "[smgpu] edge tiles: n=%d LDS=%zu B (maxP %d maxF %d maxC %d)\n", h->etl.nTiles, h->edgeLds, ev.maxPoints, ev.maxFaces, ev.maxCells);
                    if (h->edgeTilesOk && envInt("SMGPU_SIDE_STREAM", sideStreamDefault(t.nPoints))) {
                        if (depInit(h)) return cleanup(1);
                        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess)
                            return cleanup(fail("side stream creation failed"));
                    }
                }
            }
            if (rc) return cleanup(1);
            {   // the largest footprint of one tile (geomLds lays the arrays out by the tile's own counts)
                size_t mx = 0;
                for (int ti = 0; ti < h->gt.nTiles; ++ti)
                    mx = std::max(mx, 3 * (size_t)(h->gt.tpOff[(size_t)ti + 1] - h->gt.tpOff[(size_t)ti]) +
                                          (size_t)(SMGPU_GEOM_AOS ? kGF : 6) * (size_t)(h->gt.tfOff[(size_t)ti + 1] - h->gt.tfOff[(size_t)ti]));
                h->geomLds = sizeof(double) * mx;
            }
            h->smoothLds = sizeof(double) * 3 * maxTileTotal(h->stl.nTiles, {&h->stl.tcOff, &h->stl.tnOff});
            if (envInt("SMGPU_VERBOSE", 0))
                std::fprintf(stderr, "[smgpu] tiles: geom T=%d n=%d LDS=%zu B (maxP %d maxF %d; staged faces x%.3f, points x%.3f of the mesh's)  smooth T=%d n=%d LDS=%zu B (maxC %d maxN %d)\n",
                             h->geomT, h->gt.nTiles, h->geomLds, g.maxPoints, g.maxFaces, (double)h->gt.tfOff.back() / std::max(1, t.nFaces),
                             (double)h->gt.tpOff.back() / std::max(1, t.nPoints), h->smoothT, h->stl.nTiles, h->smoothLds, v.maxCells, v.maxPoints);
        }
    }
    {
        const char* ea = std::getenv("SMGPU_EDGE_ANGLE");
        h->eaCoop = !(ea && std::string(ea) == "faithful");
        for (int p0 = 0; p0 < t.nPoints; p0 += kEaPointsPerBlock) {
            const int p1 = std::min(t.nPoints, p0 + kEaPointsPerBlock);
            h->eaMaxEntries = std::max(h->eaMaxEntries, t.pointEdges.off[p1] - t.pointEdges.off[p0]);
        }
        if (sizeof(double) * 9 * (size_t)h->eaMaxEntries > 60 * 1024) h->eaCoop = false;   // extreme valences: per-point form
    }
    lapC("tile tables adopted / uploaded");
    State& s = h->st;
    const size_t P = t.nPoints, C = t.nCells, F = t.nFaces, E = t.nEdges;
    rc |= devAlloc(h, &h->bufA, 3 * P);
    rc |= devAlloc(h, &h->bufB, 3 * P);
    rc |= devAlloc(h, &s.prop, 3 * P);
    rc |= devAlloc(h, &h->dStepSqr, P + 4);
    rc |= devAlloc(h, &s.fCtr, 3 * F);
    rc |= devAlloc(h, &s.fArea, 3 * F);
    s.avgPacked = (h->useTiles && h->edgeTilesOk && h->avgPackedCount) ? 1 : 0;
    rc |= devAlloc(h, &s.fAvg, 3 * (s.avgPacked ? h->avgPackedCount : F));
    rc |= devAlloc(h, &s.cellCtr, 3 * C);
    rc |= devAlloc(h, &s.frozen, P);
    rc |= devAlloc(h, &s.edgeMin, E);
    rc |= devAlloc(h, &s.edgeMax, E);
    rc |= devAlloc(h, &s.ptMin, P);
    rc |= devAlloc(h, &s.ptMax, P);
    rc |= devAlloc(h, &s.faActive, P + 16);    // + 16: the compaction kernels read the marks 16 bytes at a time (chunkMarks)
    rc |= devAlloc(h, &h->dFaMaybe, P + 16);
    rc |= devAlloc(h, &s.faEdgeList, E);
    rc |= devAlloc(h, &s.faPointList, P);
    rc |= devAlloc(h, &h->dEaMaybe, P);
    rc |= devAlloc(h, &s.faS, P);
    rc |= devAlloc(h, &s.faN, t.pointPoints.size());
    rc |= devAlloc(h, &s.walkStack, P + 64);
    rc |= devAlloc(h, &s.acc, 1);
    rc |= devAlloc(h, &s.nearTotal, 4);
    {
        // + the boundary points' partials when k_bnd_fix finishes them (smgpu_set_boundary_smoothing)
        const size_t nPart = (size_t)std::max(gridFor(t.nPoints), h->useTiles ? h->stl.nTiles : 0) + 2 * (size_t)gridFor(t.nPoints) + 1;
        rc |= devAlloc(h, &s.blkMax, nPart);
        rc |= devAlloc(h, &s.blkCnt, nPart);
    }
    if (rc) return cleanup(1);
    lapC("work arrays allocated");
    if (hipMemset(s.acc, 0, sizeof(Accum)) != hipSuccess) return cleanup(fail("hipMemset failed"));
    if (hipMemsetAsync(s.nearTotal, 0, 4 * sizeof(unsigned long long), h->stream) != hipSuccess) return cleanup(fail("hipMemset failed"));
    if (hipMemset(s.frozen, 0, P) != hipSuccess) return cleanup(fail("hipMemset failed"));
    if (hipMemcpy(h->bufA, d->points, 3 * P * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail("upload of points failed"));
    if (hipMemcpy(s.prop, h->bufA, 3 * P * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess)
        return cleanup(fail("upload of points failed"));
    lapC("points uploaded");
    s.ptsCur = h->bufA;
    s.ptsNext = h->bufB;
    s.stats = nullptr;
    s.sharedSlot = nullptr;
    s.combA = nullptr;
    {   // the word through which the GPU tells the host how busy the face-angle walk is (updateWalkMode)
        void* dp = nullptr;
        if (hipHostMalloc((void**)&h->nActiveHost, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dp, h->nActiveHost, 0) == hipSuccess) {
            *(volatile int*)h->nActiveHost = -1;
            s.nActiveHost = (int*)dp;
        } else {
            (void)hipGetLastError();
            if (h->nActiveHost) (void)hipHostFree(h->nActiveHost);
            h->nActiveHost = nullptr;
            s.nActiveHost = nullptr;      // the replay form then stays as first decided
        }
    }
    computeAlgoBytes(h);
    if (fEdge.valid()) fEdge.wait();
    freeOwnerNeighbour(devTopo);          // (meshes without tile tables come past here with the two arrays still held)
    h->devLists.owner = h->devLists.neighbour = DeviceTopologyArrays::Arr();     // the handle's copy never owns them
    if (envInt("SMGPU_VERBOSE", 0)) std::fprintf(stderr, "[smgpu] set-up: total %.2f s\n", sinceCreate());
    *out = h;
    return 0;
}

int smgpu_destroy(smgpu_handle* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& p : h->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : h->freeEvents) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->stDev.valid)
        for (const SmoothTilesDev::Arr* a : {&h->stDev.order, &h->stDev.ptBeg, &h->stDev.tcIds, &h->stDev.tnIds, &h->stDev.selfLoc, &h->stDev.pcEll, &h->stDev.ppEll, &h->stDev.pairEll, &h->stDev.pfEll, &h->stDev.meta})
            if (a->p) (void)hipFree(a->p);
    if (h->etDev.valid)
        for (const EdgeTilesDev::Arr* a : {&h->etDev.order, &h->etDev.edgeBeg, &h->etDev.tpIds, &h->etDev.tfIds, &h->etDev.tcIds, &h->etDev.epLoc, &h->etDev.efEll, &h->etDev.ecEll, &h->etDev.meta})
            if (a->p) (void)hipFree(a->p);
    if (h->gtDev.valid)      // (a create that failed before the handle took the device-built tile tables over)
        for (const GeomTilesDev::Arr* a : {&h->gtDev.cellOrder, &h->gtDev.cellBeg, &h->gtDev.tpIds, &h->gtDev.tfIds, &h->gtDev.faceVerts, &h->gtDev.cellFaces, &h->gtDev.meta})
            if (a->p) (void)hipFree(a->p);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->nActiveHost) (void)hipHostFree(h->nActiveHost);
    for (hipEvent_t e : h->evWalkLag) if (e) (void)hipEventDestroy(e);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->evToExch) (void)hipEventDestroy(h->evToExch);
    if (h->evFromExch) (void)hipEventDestroy(h->evFromExch);
    if (h->depWords) (void)hipFree(h->depWords);
    if (h->bndSide) { (void)hipStreamSynchronize(h->bndSide); (void)hipStreamDestroy(h->bndSide); }
    if (h->evBndFork) (void)hipEventDestroy(h->evBndFork);
    if (h->evBndJoin) (void)hipEventDestroy(h->evBndJoin);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int smgpu_get_sizes(smgpu_handle* h, smgpu_sizes* o) {
    if (!h || !o) return fail("null argument");
    const Topology& t = h->topo;
    o->nPoints = t.nPoints; o->nCells = t.nCells; o->nFaces = t.nFaces; o->nInternalFaces = t.nInternalFaces; o->nEdges = t.nEdges;
    o->nnzFacePoints = t.facePoints.nnz(); o->nnzPointCells = t.pointCells.nnz(); o->nnzPointPoints = (int64_t)t.pointPoints.size();
    o->nnzPointFaces = t.facePoints.nnz(); o->nnzEdgeFaces = t.edgeFaces.nnz(); o->nnzEdgeCells = t.edgeCells.nnz();
    o->nnzCellFaces = t.cellFacesGeom.nnz();
    o->deviceBytes = h->deviceBytes;
    return 0;
}

int smgpu_mesh_stats(smgpu_handle* h, double* minEdge, double* maxEdge) {
    if (!h) return fail("null handle");
    HIP_OK(hipSetDevice(h->device));
    unsigned long long* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, 16));
    struct Release { void* p; ~Release() { if (p) (void)hipFree(p); } } release{d};
    const double big = 1.0e300, zero = 0.0;
    unsigned long long init[2];
    std::memcpy(&init[0], &big, 8);
    std::memcpy(&init[1], &zero, 8);
    HIP_OK(hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_edge_stats, dim3(std::max(1, std::min(gridFor(h->mv.nEdges), 1024))), dim3(kBlock), 0, h->stream, h->mv, h->st.ptsCur, d, d + 1);
    unsigned long long out[2];
    HIP_OK(hipMemcpyAsync(out, d, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    std::memcpy(minEdge, &out[0], 8);
    std::memcpy(maxEdge, &out[1], 8);
    return 0;
}

namespace smgpu {
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// argument j of case i: by case class random moderate values, raw bit patterns, specials, values at the range tests' edges
__device__ __forceinline__ double selftestArg(uint64_t seed, long i, int j) {
    const uint64_t r = mix64(seed + 8ull * (uint64_t)i + (uint64_t)j);
    const uint64_t mant = r & 0xFFFFFFFFFFFFFull, sign = r >> 63;
    const unsigned pick = (unsigned)((r >> 52) & 0x7ff);
    uint64_t expo;
    switch ((i + (j == 3 ? 0 : 0)) & 7) {
        case 0: case 1: case 2: expo = 1023 - 40 + pick % 80; break;                     // moderate
        case 3: return __longlong_as_double((long long)r);                                // any bit pattern
        case 4: {                                                                          // zeros, denormals, inf, nan, tiny
            const unsigned k = pick % 6;
            if (k == 0) return sign ? -0.0 : 0.0;
            if (k == 1) return __longlong_as_double((long long)((sign << 63) | mant));    // denormal
            if (k == 2) return __longlong_as_double((long long)((sign << 63) | (0x7ffull << 52)));
            if (k == 3) return __longlong_as_double((long long)((0x7ffull << 52) | mant | 1));
            expo = 1 + pick % 60; break;
        }
        case 5: {                                                                          // the edges of the range tests
            const unsigned edges[6] = {1023 - 767, 1023 - 250, 1023 + 250, 2046, 1, 1023 - 969};
            expo = edges[pick % 6] + (pick / 6) % 5 - 2;
            if ((long long)expo < 1) expo = 1;
            if (expo > 2046) expo = 2046;
            break;
        }
        case 6: expo = 1 + pick % 2046; break;                                             // any exponent
        default: expo = 1023 - 300 + pick % 600; break;
    }
    const uint64_t sgn = (j == 0 && (i & 8)) ? 0 : sign;                                   // the sqrt argument: mostly positive
    return __longlong_as_double((long long)((sgn << 63) | (expo << 52) | mant));
}
__device__ __forceinline__ bool sameBits(double a, double b) { return (a != a && b != b) || __double_as_longlong(a) == __double_as_longlong(b); }
__global__ void k_selftest_fpexact(uint64_t seed, long n, unsigned long long* bad) {
    unsigned long long mine = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double x = selftestArg(seed, i, 0), y = selftestArg(seed, i, 1), z = selftestArg(seed, i, 2), w = selftestArg(seed, i, 3);
        if (!sameBits(sqrtExact(x), sqrt(x))) ++mine;
        double a0, a1, a2, a3;
        sqrtExact4(fabs(x), fabs(y), z, fabs(w), a0, a1, a2, a3);
        if (!sameBits(a0, sqrt(fabs(x))) || !sameBits(a1, sqrt(fabs(y))) || !sameBits(a2, sqrt(z)) || !sameBits(a3, sqrt(fabs(w)))) ++mine;
        const V3 q = divExact(v3(x, y, z), w);
        if (!sameBits(q.x, x / w) || !sameBits(q.y, y / w) || !sameBits(q.z, z / w)) ++mine;
        const V3 q6 = divExact(v3(x, y, z), 6.0);
        if (!sameBits(q6.x, x / 6.0) || !sameBits(q6.y, y / 6.0) || !sameBits(q6.z, z / 6.0)) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}
}  // namespace smgpu

int smgpu_debug_selftest_fpexact(int32_t device, uint64_t seed, int64_t n, int64_t* mismatches) {
    if (!mismatches || n < 0) return fail("smgpu_debug_selftest_fpexact: bad arguments");
    if (hipSetDevice(device) != hipSuccess) return fail("smgpu_debug_selftest_fpexact: hipSetDevice failed");
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, sizeof(*d)) != hipSuccess) return fail("smgpu_debug_selftest_fpexact: hipMalloc failed");
    unsigned long long host = 0;
    bool ok = hipMemset(d, 0, sizeof(*d)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(smgpu::k_selftest_fpexact, dim3(2048), dim3(256), 0, 0, seed, (long)n, d);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(&host, d, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) return fail("smgpu_debug_selftest_fpexact: launch failed");
    *mismatches = (int64_t)host;
    return 0;
}

int smgpu_set_foam_variant(smgpu_handle* h, int32_t variant) {
    if (!h) return fail("null handle");
    if (variant != SMGPU_FOAM_COM && variant != SMGPU_FOAM_ORG) return fail("smgpu_set_foam_variant: unknown variant");
    h->foamOrg = variant == SMGPU_FOAM_ORG;
    h->geomAheadDone = false;
    return 0;
}

int smgpu_set_sync_variant(smgpu_handle* h, int32_t variant) {
    if (!h) return fail("null handle");
    if (variant != SMGPU_SYNC_MASTER && variant != SMGPU_SYNC_OWN) return fail("smgpu_set_sync_variant: unknown variant");
    h->st.ownFold = variant == SMGPU_SYNC_OWN ? 1 : 0;
    return 0;
}

int smgpu_set_params(smgpu_handle* h, const smgpu_params* p) {
    if (!h || !p) return fail("null argument");
    if (!(p->maxStepLength > 0.0)) return fail("maxStepLength must be > 0");
    h->prm = *p;
    h->prmSet = true;
    h->walkMode = -1;
    h->walkSwitches = 0;
    h->walkDecisions = 0;
    if (h->nActiveHost) *(volatile int*)h->nActiveHost = -1;
    h->geomAheadDone = false;
    computeAlgoBytes(h);
    return 0;
}

extern "C++" {
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device), not of a handle: several handles with
// different tile capacities (SMGPU_*_CAP*), or on several devices, share it.  Raise it to the largest request seen so far.
template <typename K>
static void ensureDynLds(K kernel, int device, size_t bytes) {
    static std::mutex mu;
    static size_t have[64] = {0};
    if (bytes <= 64 * 1024 || device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(mu);
    if (have[device] >= bytes) return;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) have[device] = bytes;
    else (void)hipGetLastError();
}
template <int T, bool ORG>
static void launchGeomTileAs(smgpu_handle* h, const MeshView& m, const State& s, int wantAvg, const int* tileList, int nTiles, hipEvent_t evA, hipEvent_t evB,
                             bool withBndPre) {
    if (withBndPre) {   // the boundary pre-kernels in the first workgroups of this launch (k_geom_tile_bnd)
        const int nNormal = (h->bv.nB + T - 1) / T, nFeat = (h->bv.nFeat + (T / 64) - 1) / (T / 64);
        const int nBnd = ((nNormal + nFeat + 7) / 8) * 8;
        ensureDynLds(k_geom_tile_bnd<T, ORG>, h->device, h->geomLds);
        hipExtLaunchKernelGGL((k_geom_tile_bnd<T, ORG>), dim3(nBnd + tileGrid(nTiles, h->xcdMap)), dim3(T), (uint32_t)h->geomLds, h->stream, evA, evB, 0, m, s, h->gv,
                              wantAvg, h->writeFaces ? 1 : 0, tileList, nTiles, h->xcdMap, h->deferN, h->deferIter, h->deferLocal, h->deferHist, h->bv, nBnd, nNormal);
    } else {
        ensureDynLds(k_geom_tile<T, ORG>, h->device, h->geomLds);
        hipExtLaunchKernelGGL((k_geom_tile<T, ORG>), dim3(tileGrid(nTiles, h->xcdMap)), dim3(T), (uint32_t)h->geomLds, h->stream, evA, evB, 0, m, s, h->gv, wantAvg,
                              h->writeFaces ? 1 : 0, tileList, nTiles, h->xcdMap, h->deferN, h->deferIter, h->deferLocal, h->deferHist);
    }
    h->deferN = 0;
    h->deferLocal = h->deferHist = nullptr;
}
template <int T>
static void launchGeomTile(smgpu_handle* h, const MeshView& m, const State& s, int wantAvg, const int* tileList, int nTiles, hipEvent_t evA, hipEvent_t evB,
                           bool withBndPre) {
    if (h->foamOrg) launchGeomTileAs<T, true>(h, m, s, wantAvg, tileList, nTiles, evA, evB, withBndPre);
    else launchGeomTileAs<T, false>(h, m, s, wantAvg, tileList, nTiles, evA, evB, withBndPre);
}
template <bool FINAL, int T>
static void launchSmoothTile(smgpu_handle* h, const MeshView& m, const State& s, const Prm& prm, const int* tileList, int nTiles, hipEvent_t evA, hipEvent_t evB) {
    ensureDynLds(k_smooth_tile<FINAL, T>, h->device, h->smoothLds);
    hipExtLaunchKernelGGL((k_smooth_tile<FINAL, T>), dim3(tileGrid(nTiles, h->xcdMap)), dim3(T), (uint32_t)h->smoothLds, h->stream, evA, evB, 0, m, s, prm, h->sv,
                          tileList, nTiles, h->xcdMap);
}
template <bool FINAL>
static int launchBndFix(smgpu_handle* h, int partialBase) {
    if (h->bv.nB == 0) return 0;   // a sub-domain without boundary points (interior rank of a decomposition)
    return launchK(h, K_BND, [&] {
        hipLaunchKernelGGL(k_bnd_fix<FINAL>, dim3(gridFor(2 * (int64_t)h->bv.nB)), dim3(kBlock), 0, h->stream, h->mv, h->st, makePrm(h), h->bv, partialBase);
    });
}
static int launchBndPre(smgpu_handle* h, const MeshView& m, const State& s, hipStream_t stream) {
    if (h->bv.nB == 0) return 0;
    return launchK(h, K_BND, [&] {
        hipLaunchKernelGGL(k_bnd_normals, dim3(gridFor(h->bv.nB)), dim3(kBlock), 0, stream, m, s, h->bv);
        if (h->bv.nFeat) hipLaunchKernelGGL(k_bnd_feature, dim3(h->bv.nFeat), dim3(64), 0, stream, m, s, h->bv);
    }, stream);
}
// Start of an iteration with boundary point smoothing: both kernels only read the current coordinates, so they run on
// a side stream while the geometry kernel has the main one; runSmooth joins.
static int runBndPre(smgpu_handle* h) {
    if (h->bndOn && h->bndInGeom && h->useTiles && h->bv.nB > 0) return 0;   // they ride in the geometry launch (k_geom_tile_bnd)
    if (!h->bndOn || !h->bndSide || h->bndPreInFlight || h->timing) return 0;   // (timing passes: every kernel alone, see forkFaFilter)
    if (depSignal(h, DEP_BND_FORK, h->stream, h->evBndFork) || depWait(h, DEP_BND_FORK, h->bndSide, h->evBndFork)) return 1;
    if (launchBndPre(h, h->mv, h->st, h->bndSide)) return 1;
    if (depSignal(h, DEP_BND_JOIN, h->bndSide, h->evBndJoin)) return 1;
    h->bndPreInFlight = true;
    return 0;
}
// tileList == NULL: all tiles; otherwise the nTiles listed ones (multi-rank interior / shared split)
template <bool FINAL>
static int runSmooth(smgpu_handle* h, const MeshView& m, const State& s, const Prm& prm, const int* tileList = nullptr, int nList = 0) {
    const int kid = FINAL ? K_SMOOTH_FINAL : K_SMOOTH_PROP;
    // boundary point smoothing: normals and feature edge projections of the current coordinates first (SM.C:2266,
    // BPS.C:866) -- started next to the geometry kernel by runBndPre when there is a side stream; after the smoothing
    // kernel k_bnd_fix finishes the boundary points it skipped
    const bool withBnd = h->bndOn && !h->haloOn;   // with a halo smgpu_iter_begin / smgpu_iter_mid place the boundary kernels
    if (withBnd) {
        if (h->bndPreDone) h->bndPreDone = false;
        else if (h->bndPreInFlight) {
            if (depWait(h, DEP_BND_JOIN, h->stream, h->evBndJoin)) return 1;
            h->bndPreInFlight = false;
        } else if (launchBndPre(h, m, s, h->stream)) return 1;
    }
    if (h->useTiles) {
        const int nT = tileList ? nList : h->stl.nTiles;
        if (nT == 0) return 0;
        if (launchKDispatch(h, kid, [&](hipEvent_t evA, hipEvent_t evB) {
                if (h->smoothT == 64) launchSmoothTile<FINAL, 64>(h, m, s, prm, tileList, nT, evA, evB);
                else if (h->smoothT == 128) launchSmoothTile<FINAL, 128>(h, m, s, prm, tileList, nT, evA, evB);
                else launchSmoothTile<FINAL, 256>(h, m, s, prm, tileList, nT, evA, evB);
            })) return 1;
    } else if (launchK(h, kid, [&] { hipLaunchKernelGGL(k_smooth<FINAL>, dim3(gridFor(m.nPoints)), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
    if (withBnd) {
        const int base = h->useTiles ? h->stl.nTiles : gridFor(m.nPoints);
        return launchK(h, K_BND, [&] { hipLaunchKernelGGL(k_bnd_fix<FINAL>, dim3(gridFor(2 * (int64_t)h->bv.nB)), dim3(kBlock), 0, h->stream, m, s, prm, h->bv, base); });
    }
    return 0;
}

}  // extern "C++"

// geometry of the current coordinates: OpenFOAM face centres/areas + cell centres.
// tileList != NULL restricts the tiled form to the listed tiles; fromNext reads the coordinates of the
// iteration being finished (ptsNext) instead of ptsCur (multi-rank look-ahead, smgpu_iter_ahead).
// withBndPre: this is the geometry launch at the start of an iteration with boundary point smoothing -- the normals / feature
// projection kernels of the current coordinates ride in it (bndPreDone tells runSmooth / smgpu_iter_begin)
static int runGeometry(smgpu_handle* h, const int* tileList = nullptr, int nList = 0, bool fromNext = false, bool withBndPre = false) {
    const MeshView& m = h->mv;
    State s = h->st;
    if (fromNext) s.ptsCur = h->st.ptsNext;
    const int wantAvg = h->prm.faceAngleConstraint ? 1 : 0;
    if (h->useTiles) {
        const int nT = tileList ? nList : h->gt.nTiles;
        if (nT == 0) return 0;
        const bool bnd = withBndPre && h->bndOn && h->bndInGeom && !fromNext && h->bv.nB > 0 && !h->bndPreInFlight && !h->bndPreDone;
        if (bnd) h->bndPreDone = true;
        return launchKDispatch(h, K_GEOM_TILE, [&](hipEvent_t evA, hipEvent_t evB) {
            if (h->geomT == 64) launchGeomTile<64>(h, m, s, wantAvg, tileList, nT, evA, evB, bnd);
            else if (h->geomT == 128) launchGeomTile<128>(h, m, s, wantAvg, tileList, nT, evA, evB, bnd);
            else launchGeomTile<256>(h, m, s, wantAvg, tileList, nT, evA, evB, bnd);
        });
    }
    if (launchK(h, K_FACE_GEOM, [&] { hipLaunchKernelGGL(k_face_geom, dim3(gridFor(m.nFaces)), dim3(kBlock), 0, h->stream, m, s, wantAvg, h->foamOrg ? 1 : 0); })) return 1;
    if (launchK(h, K_CELL_CENTRES, [&] { hipLaunchKernelGGL(k_cell_centres, dim3(gridFor(m.nCells)), dim3(kBlock), 0, h->stream, m, s, h->foamOrg ? 1 : 0); })) return 1;
    return 0;
}

// ---- face-angle walk with host replay (kernels_walk.hpp) ---------------------------------------
static int ensureWalkBuffers(smgpu_handle* h) {
    if (h->walkAlloc) return 0;
    const Topology& t = h->topo;
    if (envInt("SMGPU_WALK_MEMO_STATS", 0) && !SMGPU_WALK_MEMO) return fail("SMGPU_WALK_MEMO_STATS needs a build with -DSMGPU_WALK_MEMO=1 (make HIPFLAGS+=-DSMGPU_WALK_MEMO=1)");
    if (envInt("SMGPU_WALK_MEMO_STATS", 0) && !h->dWalkMemo) {
        if (devAlloc(h, &h->dWalkMemo, (size_t)t.nPoints + 2)) return 1;
        HIP_OK(zeroNow(h->dWalkMemo, 0, ((size_t)t.nPoints + 2) * sizeof(unsigned long long)));
    }
    const size_t P = t.nPoints, E = t.pointPoints.size();
    h->walkBlocks = gridFor(t.nPoints);
    WalkView& w = h->wv;
    int rc = 0;
    rc |= devAlloc(h, &w.activeSlot, P);
    rc |= devAlloc(h, &w.blkA, (size_t)h->walkBlocks + 1);
    rc |= devAlloc(h, &w.blkE, (size_t)h->walkBlocks + 1);
    rc |= devAlloc(h, &w.header, 4);
    rc |= devAlloc(h, &w.actIds, P);
    rc |= devAlloc(h, &w.actEntOff, P + 1);
    rc |= devAlloc(h, &w.actBits, P);
    rc |= devAlloc(h, &w.entOwner, E + P);
    rc |= devAlloc(h, &w.entNbr, E);
    rc |= devAlloc(h, &w.entSlot, E);
    rc |= devAlloc(h, &w.entBits, E);
    rc |= devAlloc(h, &w.relSlot, P);
    rc |= devAlloc(h, &w.header2, 4);
    rc |= devAlloc(h, &w.relBits, P);
    rc |= devAlloc(h, &w.hdrPos, P);
    rc |= devAlloc(h, &w.items, E + P);
    rc |= devAlloc(h, &h->dWalkOps, 64);
    if (rc) return 1;
    HIP_OK(zeroNow(h->dWalkOps, 0, 64 * sizeof(unsigned long long)));
    // the stars' static records (kernels_walk.hpp: StarCache): a pool for an eighth of the points (the points outside the good range
    // are a few per cent of a mesh worth smoothing; records beyond the pool are staged from the addressing every iteration, as before)
    if (h->walkStar && h->walkPack && h->walkCache && !SMGPU_WALK_MEMO) {
        StarCache& c = h->starCache;
        c.capacity = std::max(1, envInt("SMGPU_WALK_CACHE_CAP", (int)std::max<size_t>(131072, P / 8)));
        // the pool is an accelerator, not a necessity (112 bytes per mesh point by default): a device that cannot hold it runs every
        // star through k_walk_pred_pack as with SMGPU_WALK_CACHE=0; a smaller pool is tried first
        void* pool = nullptr;
        while (hipMalloc(&pool, (size_t)c.capacity * sizeof(StarRec)) != hipSuccess) {
            (void)hipGetLastError();
            pool = nullptr;
            if (c.capacity <= 4096) break;
            c.capacity /= 4;
        }
        if (!pool) {
            if (envInt("SMGPU_VERBOSE", 0) >= 1) std::fprintf(stderr, "[smgpu] no device memory for the stars' records: every star is staged from the addressing\n");
            c = StarCache{nullptr, nullptr, nullptr, 0};
            h->walkAlloc = true;
            return 0;
        }
        h->allocs.push_back(pool);
        h->deviceBytes += (int64_t)((size_t)c.capacity * sizeof(StarRec));
        c.pool = (StarRec*)pool;
        if (devAlloc(h, &c.slot, P) || devAlloc(h, &c.count, 4)) return 1;
        // (on the engine's stream: it is a non-blocking stream, which a hipMemset on the null stream does not order with -- with several
        // processes on one device the first walk's kernels overtook the fill and read slots that were not -1 yet)
        HIP_OK(hipMemsetAsync(c.slot, 0xFF, P * sizeof(int), h->stream));      // -1: no record yet
        HIP_OK(hipMemsetAsync(c.count, 0, 4 * sizeof(int), h->stream));
    }
    h->walkAlloc = true;
    return 0;
}

static int ensurePinned(smgpu_handle* h, size_t bytes) {
    if (h->pinnedBytes >= bytes) return 0;
    if (h->pinned) HIP_OK(hipHostFree(h->pinned));
    h->pinned = nullptr;
    h->pinnedBytes = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    HIP_OK(hipHostMalloc(&h->pinned, want, hipHostMallocDefault));
    h->pinnedBytes = want;
    return 0;
}

// The reference's stack walk SM.C:1347-1434 over the item sequence of k_rel_fill.  Slots ascend with the
// point id, so the reference's pop order (highest id first) is the order of the items; a point frozen by a
// neighbour is pushed and re-visited before the walk continues (SM.C:1431).  Non-active points never act
// (SM.C:1367-1369), so only freezing them is recorded.
//
// The first visits are one flat loop over the items without data-dependent branches (the decisions are close to
// 50/50 and the per-point entry lists are 2-3 long: the nested, branchy form spent its time in mispredictions):
// candidates are always stored and the cursors advance by the 0/1 outcome.  Re-visits are rare and take the
// ordinary path (drain) before the next point starts.
static void drainWalk(const WalkItem* __restrict items, int N, uint8_t* __restrict frozen, int* __restrict stack, int* __restrict out,
                      int* spIO, size_t* nOutIO) {
    int sp = *spIO;
    size_t nOut = *nOutIO;
    while (sp) {
        int i = stack[--sp];                                                     // header position
        const WalkItem hd = items[i];
        unsigned sel = 2;                                                        // SM.C:1376-1385
        if (!frozen[hd.slot]) {
            if (hd.bits & 3u) { frozen[hd.slot] = 1; out[nOut++] = hd.id; }      // SM.C:1391-1399
            else if (hd.bits & 64u) sel = 1;
        }
        for (++i; i < N && !(items[i].bits & 0x80u); ++i) {                      // SM.C:1406-1433
            const WalkItem e = items[i];
            if (frozen[e.slot] || !(e.bits & sel)) continue;                     // neighbour already frozen / not hurt
            out[nOut++] = e.id;
            if (e.bits & 32u) frozen[e.slot] = 1;
            if (e.bits & 16u) { stack[sp++] = e.hpos; __builtin_prefetch(&items[e.hpos]); }
        }
    }
    *spIO = sp;
    *nOutIO = nOut;
}

static void replayWalk(const WalkItem* __restrict items, int N, int nA, const uint8_t* __restrict relBits, std::vector<uint8_t>& frozenV,
                       std::vector<int>& stackV, std::vector<int>& outV) {
    frozenV.resize((size_t)nA + 2);
    stackV.resize((size_t)nA + 8);
    if (outV.size() < (size_t)2 * N + 8) outV.resize((size_t)2 * N + 8);
    uint8_t* __restrict frozen = frozenV.data();
    int* __restrict stack = stackV.data();
    int* __restrict out = outV.data();
    for (int a = 0; a < nA; ++a) frozen[a] = (relBits[a] >> 2) & 1;
    frozen[nA] = 0;                                  // pseudo slots of the sinks: not frozen / frozen before the walk
    frozen[nA + 1] = 1;
    size_t nOut = 0;
    int sp = 0;
    unsigned sel = 2;
    // first visits: one pass over the items, no data-dependent branch besides the (rare) drain
    for (int i = 0; i < N; ++i) {
        const WalkItem it = items[i];
        const unsigned bits = it.bits;
        const unsigned isH = bits >> 7;
        if (__builtin_expect((isH * (unsigned)sp) != 0, 0)) {   // the previous point is complete: its pushes come first
            int spT = sp;
            size_t nT = nOut;
            drainWalk(items, N, frozen, stack, out, &spT, &nT);
            sp = spT;
            nOut = nT;
        }
        const unsigned nf = frozen[it.slot] ^ 1u;
        const unsigned act = nf & (0u - (unsigned)((bits & sel) != 0));
        out[nOut] = it.id;
        nOut += act;
        frozen[it.slot] |= (uint8_t)(act & (bits >> 5));
        stack[sp] = it.hpos;
        sp += (int)(act & (bits >> 4));
        const unsigned selH = 2u - (nf & (bits >> 6));                           // header: SM.C:1376-1399
        sel ^= (sel ^ selH) & (0u - isH);
    }
    {
        int spT = sp;
        size_t nT = nOut;
        drainWalk(items, N, frozen, stack, out, &spT, &nT);
        nOut = nT;
    }
    outV.resize(nOut);
}

// Busy-wait for the stream: the host replay that follows is latency critical, and a blocking wait lets the core
// drop to a low-power state (the replay then ran ~2x slower than on a busy core).
static int spinSync(hipStream_t stream) {
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail(std::string("hipStreamQuery: ") + hipGetErrorString(e));
        __builtin_ia32_pause();
    }
}

static int runHostWalk(smgpu_handle* h) {
    if (ensureWalkBuffers(h)) return 1;
    const MeshView& m = h->mv;
    State s = h->st;
    const Prm prm = makePrm(h);
    WalkView w = h->wv;
    if (ensurePinned(h, 64)) return 1;
    const int gChunks = chunkGrid(m.nPoints);
    if (launchK(h, K_FA_PRED, [&] {
            hipLaunchKernelGGL(k_walk_count, dim3(gChunks), dim3(kBlock), 0, h->stream, m, s, w, h->starCache.pool ? h->starCache.count + 1 : (int*)nullptr);
            hipLaunchKernelGGL(k_walk_fill, dim3(gChunks), dim3(kBlock), 0, h->stream, m, s, w, h->starCache.pool ? (const int*)h->starCache.slot : (const int*)nullptr, h->starCache.pool ? h->starCache.count + 1 : (int*)nullptr);
        })) return 1;
    int* hdr = (int*)h->pinned;
    HIP_OK(hipMemcpyAsync(hdr, w.header, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (spinSync(h->stream)) return 1;
    const int nA = hdr[0], nE = hdr[1];
    if (nA <= 0) return 0;
    if (launchK(h, K_FA_PRED, [&] {
            if (h->walkStar && h->walkPack && h->starCache.pool) {
                hipLaunchKernelGGL(k_walk_star_build, dim3((nA + 7) / 8), dim3(kPackBlock), 0, h->stream, m, s, prm, w, nA, nE, h->starCache);
                hipLaunchKernelGGL(k_walk_pred_cached, dim3((nA + 7) / 8), dim3(kPackBlock), 0, h->stream, m, s, prm, w, nA, nE, h->timing ? h->dWalkOps : nullptr, h->starCache);
                hipLaunchKernelGGL(k_walk_pred_pack_rest, dim3((nA + 7) / 8), dim3(kPackBlock), 0, h->stream, m, s, prm, w, nA, nE, h->timing ? h->dWalkOps : nullptr, h->starCache);
            } else if (h->walkStar && h->walkPack) hipLaunchKernelGGL(k_walk_pred_pack, dim3((nA + 7) / 8), dim3(kPackBlock), 0, h->stream, m, s, prm, w, nA, nE, h->timing ? h->dWalkOps : nullptr, h->dWalkMemo);
            else if (h->walkStar) hipLaunchKernelGGL(k_walk_pred_star, dim3((nA + 7) / 8), dim3(kBlock), 0, h->stream, m, s, prm, w, nA, nE, h->timing ? h->dWalkOps : nullptr);
            hipLaunchKernelGGL(k_walk_pred_self, dim3(gridFor(32 * (int64_t)nA)), dim3(kBlock), 0, h->stream, m, s, prm, w, nA, nE, h->walkStar ? 1 : 0);
            hipLaunchKernelGGL(k_walk_pred, dim3(std::max(1, gridFor(32 * (int64_t)nE))), dim3(kBlock), 0, h->stream, m, s, prm, w, nA, nE, h->walkStar ? 1 : 0);
        })) return 1;
    // second compaction: only the points that can act and only their true entries go to the host
    if (launchK(h, K_FA_PRED, [&] {
            hipLaunchKernelGGL(k_rel_count, dim3(relGrid(nA)), dim3(kBlock), 0, h->stream, w);
            hipLaunchKernelGGL(k_rel_fill, dim3(relGrid(nA)), dim3(kBlock), 0, h->stream, w, FixView{});
        })) return 1;
    HIP_OK(hipMemcpyAsync(hdr, w.header2, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (spinSync(h->stream)) return 1;
    const int nR = hdr[0], nB = hdr[1];
    if (nR <= 0) return 0;
    const int nItems = nR + nB;
    if (launchK(h, K_FA_PRED, [&] { hipLaunchKernelGGL(k_rel_link, dim3(gridFor(nItems)), dim3(kBlock), 0, h->stream, w, nItems, nR, 0); })) return 1;
    const size_t oItems = 0, oRel = oItems + sizeof(WalkItem) * (size_t)nItems, total = oRel + (size_t)nR;
    if (ensurePinned(h, total + 16)) return 1;
    char* base = (char*)h->pinned;
    HIP_OK(hipMemcpyAsync(base + oItems, w.items, sizeof(WalkItem) * (size_t)nItems, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(base + oRel, w.relBits, (size_t)nR, hipMemcpyDeviceToHost, h->stream));
    if (spinSync(h->stream)) return 1;
    if (const char* dump = std::getenv("SMGPU_DUMP_WALK")) {          // offline analysis of the replay input
        static int calls = 0;
        if (++calls == envInt("SMGPU_DUMP_WALK_CALL", 3)) {
            if (FILE* f = std::fopen(dump, "wb")) {
                const int64_t hdr3[3] = {nR, nB, (int64_t)total};
                std::fwrite(hdr3, sizeof(hdr3), 1, f);
                std::fwrite(base, 1, total, f);
                std::fclose(f);
            }
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // the walk runs on a copy in ordinary memory (a streaming copy that also warms the cache)
    h->walkHost.resize(total);
    std::memcpy(h->walkHost.data(), base, total);
    const char* hb = h->walkHost.data();
    const double copyMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    replayWalk((const WalkItem*)(hb + oItems), nItems, nR, (const uint8_t*)(hb + oRel), h->walkFrozen, h->walkStack, h->walkOut);
    const double replayMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (h->timing) h->ms[K_FA_WALK] += replayMs;
    if (envInt("SMGPU_VERBOSE", 0) > 1)
        std::fprintf(stderr, "[smgpu] walk: active %d entries %d -> acting %d entries %d, froze %zu, replay %.3f ms (copy %.3f ms)\n", nA, nE, nR, nB,
                     h->walkOut.size(), replayMs, copyMs);
    h->launches[K_FA_WALK]++;
    const int nOut = (int)h->walkOut.size();
    if (nOut > 0) {
        // the id list reuses the pinned buffer's head and the (now consumed) entOwner array on the device
        std::memcpy(base, h->walkOut.data(), sizeof(int) * (size_t)nOut);
        HIP_OK(hipMemcpyAsync(w.entOwner, base, sizeof(int) * (size_t)nOut, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_walk_apply, dim3(gridFor(nOut)), dim3(kBlock), 0, h->stream, s, w.entOwner, nOut);
        if (spinSync(h->stream)) return 1;   // the pinned buffer is reused by the next iteration
    }
    return 0;
}

// The same walk without the host: compaction and predicates as above with every count left on the device (the
// launches cover the largest possible count or stride over it), then k_walk_fix solves the walk as a causal fixed point
// in one persistent launch (kernels_walk.hpp).  No copy, no synchronisation.
static int runFixWalk(smgpu_handle* h) {
    if (ensureWalkBuffers(h)) return 1;
    if (!h->fixAlloc) {
        const size_t P = (size_t)h->topo.nPoints;
        FixView& f = h->fxw;
        if (devAlloc(h, &f.T, P) || devAlloc(h, &f.act, P) || devAlloc(h, &f.bar, 16) || devAlloc(h, &f.flags, 16)) return 1;
        if (envInt("SMGPU_WALK_WARM", 1)) { if (devAlloc(h, &f.actPrev, P)) return 1; HIP_OK(zeroNow(f.actPrev, 0, P)); }
        // every workgroup of the persistent launch has to be resident at once: far fewer than the chip holds (2 x 256)
        h->walkFixBlocks = std::max(1, std::min(envInt("SMGPU_WALK_BLOCKS", std::max(8, 128 / std::max(1, h->deviceShare))), 256));
        h->walkSweeps = std::max(1, envInt("SMGPU_WALK_SWEEPS", 8));
        h->fixAlloc = true;
        if (envInt("SMGPU_WALK_STATS", 0)) { const int v[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,12345}; HIP_OK(hipMemcpy(f.flags, v, sizeof(v), hipMemcpyHostToDevice)); }
    }
    const MeshView& m = h->mv;
    State s = h->st;
    const Prm prm = makePrm(h);
    WalkView w = h->wv;
    const FixView fx = h->fxw;
    const int P = m.nPoints;
    const int64_t maxEntries = (int64_t)h->topo.pointPoints.size();
    // grids: enough workgroups to fill the chip several times over; the kernels stride over the device-side counts
    const int gItems = (int)std::min<int64_t>(((int64_t)P + maxEntries + kBlock - 1) / kBlock, 256 * 8);
    const int gChunks = chunkGrid(P), gRel = relGrid(P);
    if (launchK(h, K_FA_PRED, [&] {
            hipLaunchKernelGGL(k_walk_count, dim3(gChunks), dim3(kBlock), 0, h->stream, m, s, w, h->starCache.pool ? h->starCache.count + 1 : (int*)nullptr);
            hipLaunchKernelGGL(k_walk_fill, dim3(gChunks), dim3(kBlock), 0, h->stream, m, s, w, h->starCache.pool ? (const int*)h->starCache.slot : (const int*)nullptr, h->starCache.pool ? h->starCache.count + 1 : (int*)nullptr);
            if (h->walkStar && h->walkPack && h->starCache.pool) {
                // (the two launches around the cached one leave at once in an iteration that has nothing for them: small grids)
                hipLaunchKernelGGL(k_walk_star_build, dim3(std::min(h->starBlocks, 2048)), dim3(kPackBlock), 0, h->stream, m, s, prm, w, -1, -1, h->starCache);
                hipLaunchKernelGGL(k_walk_pred_cached, dim3(h->starBlocks), dim3(kPackBlock), 0, h->stream, m, s, prm, w, -1, -1, h->timing ? h->dWalkOps : nullptr, h->starCache);
                hipLaunchKernelGGL(k_walk_pred_pack_rest, dim3(std::min(h->starBlocks, 2048)), dim3(kPackBlock), 0, h->stream, m, s, prm, w, -1, -1, h->timing ? h->dWalkOps : nullptr, h->starCache);
            } else if (h->walkStar && h->walkPack) hipLaunchKernelGGL(k_walk_pred_pack, dim3(h->starBlocks), dim3(kPackBlock), 0, h->stream, m, s, prm, w, -1, -1, h->timing ? h->dWalkOps : nullptr, h->dWalkMemo);
            else if (h->walkStar) hipLaunchKernelGGL(k_walk_pred_star, dim3(h->starBlocks), dim3(kBlock), 0, h->stream, m, s, prm, w, -1, -1, h->timing ? h->dWalkOps : nullptr);
            hipLaunchKernelGGL(k_walk_pred_self, dim3(h->walkStar ? 256 * 4 : 256 * 32), dim3(kBlock), 0, h->stream, m, s, prm, w, -1, -1, h->walkStar ? 1 : 0);
            hipLaunchKernelGGL(k_walk_pred, dim3(h->walkStar ? 256 * 8 : 256 * 64), dim3(kBlock), 0, h->stream, m, s, prm, w, -1, -1, h->walkStar ? 1 : 0);
        })) return 1;
    if (launchK(h, K_FA_WALK, [&] {
            hipLaunchKernelGGL(k_rel_count, dim3(gRel), dim3(kBlock), 0, h->stream, w);
            hipLaunchKernelGGL(k_rel_fill, dim3(gRel), dim3(kBlock), 0, h->stream, w, fx);
            hipLaunchKernelGGL(k_rel_link, dim3(gItems), dim3(kBlock), 0, h->stream, w, -1, -1, 1);
            hipLaunchKernelGGL(k_walk_fix, dim3(h->walkFixBlocks), dim3(kFixBlock), 0, h->stream, w, fx, s, h->walkSweeps, envInt("SMGPU_WALK_LOCAL", 1));
        })) return 1;
    return 0;
}

// New tag for this iteration's faMaybe / faActive marks; the arrays are cleared only when the 8-bit tag wraps (every 255
// iterations) instead of by two P-byte fills per iteration.
static int nextFaGen(smgpu_handle* h, hipStream_t stream) {
    if (h->st.faGen == 255 || h->st.faGen == 0) {
        HIP_OK(hipMemsetAsync(h->dFaMaybe, 0, (size_t)h->mv.nPoints, stream));
        HIP_OK(hipMemsetAsync(h->st.faActive, 0, (size_t)h->mv.nPoints, stream));
        h->st.faGen = 0;
    }
    ++h->st.faGen;
    return 0;
}

static int ensureWalkBuffers(smgpu_handle* h);
// The exact face angles of the CURRENT coordinates on what the filter left open (or on everything: faMaybe == NULL) and the
// per-point minima / maxima with the good-range test (SM.C:938-975, 1367-1369).  Needs the current geometry only -- not the
// proposals, not the edge-angle results -- so it follows the filter on the side stream when there is one.
static int runFaExactPass(smgpu_handle* h, const State& s, const uint8_t* faMaybe, hipStream_t stream) {
    const MeshView& m = h->mv;
    const Prm prm = makePrm(h);
    const int gP = gridFor(m.nPoints);
    // lists pay when many points are outside the good range (the choice the walk makes once per parameter set: walkMode 0 =
    // few); on a good mesh the filter leaves nothing open and two early-exit launches are cheaper than four
    if (faMaybe && h->faLists && h->walkMode > 0) {
        // exact evaluation on the lists of what the filter left open (kernels.hpp, k_fa_collect)
        if (ensureWalkBuffers(h)) return 1;    // the block-count scratch of the walk compaction serves the listing first
        if (launchK(h, K_FA_EDGES, [&] {
                hipLaunchKernelGGL(k_fa_list_count, dim3(chunkGrid(m.nPoints)), dim3(kBlock), 0, stream, m, s, faMaybe, h->wv.blkA, h->wv.blkE);
                hipLaunchKernelGGL(k_fa_list_fill, dim3(chunkGrid(m.nPoints)), dim3(kBlock), 0, stream, m, s, faMaybe, h->wv.blkA, h->wv.blkE);
                hipLaunchKernelGGL(k_fa_edges_list, dim3(std::max(1, std::min(gridFor(m.nEdges), 256 * 32))), dim3(kBlock), 0, stream, m, s);
            }, stream)) return 1;
        return launchK(h, K_FA_POINTS, [&] { hipLaunchKernelGGL(k_fa_points_list, dim3(std::max(1, std::min(gP, 256 * 8))), dim3(kBlock), 0, stream, m, s, prm); }, stream);
    }
    if (launchK(h, K_FA_EDGES, [&] { hipLaunchKernelGGL(k_fa_edges, dim3(gridFor(m.nEdges)), dim3(kBlock), 0, stream, m, s, faMaybe); }, stream)) return 1;
    return launchK(h, K_FA_POINTS, [&] { hipLaunchKernelGGL(k_fa_points, dim3(gP), dim3(kBlock), 0, stream, m, s, prm, faMaybe); }, stream);
}

// proposal (non-final) + constraint evaluators; leaves prop / frozen on the device
// launch the face-angle filter (needs only the geometry of the current coordinates) on the side stream
static int forkFaFilter(smgpu_handle* h) {
    if (!h->side || !h->prm.faceAngleConstraint || !h->useFilter || h->exactAll || !h->edgeTilesOk || h->faFilterInFlight) return 0;
    // per-kernel timing passes run every kernel ALONE on the main stream: a duration measured while another stream's kernel
    // shares the chip is no kernel time (round 3's table priced the proposal kernel at 0.17 of the peak that way)
    if (h->timing) return 0;
    const MeshView& m = h->mv;
    const Prm prm = makePrm(h);
    if (depSignal(h, DEP_FORK, h->stream, h->evFork) || depWait(h, DEP_FORK, h->side, h->evFork)) return 1;
    if (nextFaGen(h, h->side)) return 1;
    State s = h->st;
    if (launchK(h, K_FA_FILTER, [&] {
            hipLaunchKernelGGL(k_fa_filter_tile<256>, dim3(tileGrid(h->etl.nTiles, h->xcdMap)), dim3(256), h->edgeLds, h->side, s, prm, h->ev, m.edges, h->dFaMaybe,
                               h->etl.nTiles, h->xcdMap);
        }, h->side)) return 1;
    h->faExactOnSide = h->faSideExact && h->walkMode >= 0;     // (the first pass of a parameter set decides walkMode after a sync: in order)
    if (h->faExactOnSide && runFaExactPass(h, s, h->dFaMaybe, h->side)) return 1;
    if (depSignal(h, DEP_JOIN, h->side, h->evJoin)) return 1;
    h->faFilterInFlight = true;
    return 0;
}

// The replay form of the face-angle walk (and with it the list form of the exact face-angle pass).  Few points outside the
// good range: the one-wave replay over the full flag array (two launches); many: compaction + predicates per active point +
// the fixed-point replay (nine launches that would be pure overhead on a good mesh).  SMGPU_WALK = wave | host | fix forces
// one (SMGPU_HOST_WALK = 0 | 1: the first two, as in round 1).  Otherwise the first constrained iteration of a parameter set
// reads the count once (one synchronisation, decideWalkMode), and from then on the choice follows the mesh: the
// end-of-iteration reduction leaves nActive in a pinned host word (State::nActiveHost) and every iteration is enqueued with
// the form that fits the LATEST count the GPU has published -- a mesh that starts good and degrades (or the reverse) changes
// form mid-run.  Both forms give the same result, so a stale count only costs time; the staleness is bounded by waiting, every
// 8th decision, for the iteration enqueued 8 decisions earlier (the GPU still has those 8 iterations queued: it never idles).
static int walkThreshold() { return envInt("SMGPU_HOST_WALK_THRESHOLD", 256); }
static int decideWalkMode(smgpu_handle* h) {
    const char* env = std::getenv("SMGPU_WALK");
    const char* envHost = std::getenv("SMGPU_HOST_WALK");
    h->walkForced = true;
    if (env && std::string(env) == "wave") h->walkMode = 0;
    else if (env && std::string(env) == "host") h->walkMode = 1;
    else if (env && std::string(env) == "fix") h->walkMode = 2;
    else if (envHost && std::string(envHost) != "auto") h->walkMode = std::atoi(envHost) ? 1 : 0;
    else {
        h->walkForced = false;
        Accum a;
        HIP_OK(hipMemcpyAsync(&a, h->st.acc, sizeof(Accum), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        h->walkMode = a.nActive > walkThreshold() ? 2 : 0;
    }
    return 0;
}
static int updateWalkMode(smgpu_handle* h) {
    if (!h->prm.faceAngleConstraint || h->walkMode < 0 || h->walkForced || !h->nActiveHost) return 0;
    const long n = ++h->walkDecisions;
    if ((n & 7) == 0) {
        const int cur = (int)((n >> 3) & 1);
        for (hipEvent_t& e : h->evWalkLag) if (!e) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (n >= 16) HIP_OK(hipEventSynchronize(h->evWalkLag[cur ^ 1]));   // the iteration enqueued 8 decisions ago has finished
        HIP_OK(hipEventRecord(h->evWalkLag[cur], h->stream));
    }
    const int v = *(volatile int*)h->nActiveHost;
    if (v < 0) return 0;
    // hysteresis: up above the threshold, down below half of it
    const int thr = walkThreshold();
    const int want = (h->walkMode == 2) ? (v < thr / 2 ? 0 : 2) : (v > thr ? 2 : 0);
    if (want != h->walkMode) {
        h->walkMode = want;
        ++h->walkSwitches;
        if (envInt("SMGPU_VERBOSE", 0)) std::fprintf(stderr, "[smgpu] walk: %d points outside the good range -> replay form %s\n", v, want ? "fix" : "wave");
    }
    return 0;
}

static int runConstraints(smgpu_handle* h);
static int runProposalAndConstraints(smgpu_handle* h) {
    if (!h->haloOn && updateWalkMode(h)) return 1;      // (step-wise loop: smgpu_iter_begin has decided for this iteration)
    if (forkFaFilter(h)) return 1;
    if (runSmooth<false>(h, h->mv, h->st, makePrm(h))) return 1;
    if (h->bndOn && h->haloOn && launchBndFix<false>(h, 0)) return 1;
    return runConstraints(h);
}
// the constraint evaluators on the proposals left in prop / frozen (SM.C:2361-2371)
static int runConstraints(smgpu_handle* h) {
    const MeshView& m = h->mv;
    State s = h->st;
    const Prm prm = makePrm(h);
    const int gP = gridFor(m.nPoints);
    const bool filt = h->useFilter && !h->exactAll;
    if (h->prm.edgeAngleConstraint) {
        const uint8_t* eaMaybe = nullptr;
        if (filt && h->eaCoop) {
            const float cosSmall = (float)std::cos(prm.smallAngle);
            if (h->useTiles && h->smoothT == 256) {
                const size_t ldsB = sizeof(float) * (6 * (size_t)h->sv.maxPoints + 16);
                if (launchK(h, K_EA_FILTER, [&] {
                        hipLaunchKernelGGL(k_ea_filter_tile<256>, dim3(tileGrid(h->stl.nTiles, h->xcdMap)), dim3(256), ldsB, h->stream, m, s, h->sv, cosSmall, h->dEaMaybe,
                                           h->stl.nTiles, h->xcdMap);
                    })) return 1;
            } else if (launchK(h, K_EA_FILTER, [&] { hipLaunchKernelGGL(k_edge_angle_filter, dim3(gP), dim3(kBlock), 0, h->stream, m, s, prm, cosSmall, h->dEaMaybe); })) return 1;
            eaMaybe = h->dEaMaybe;
        }
        if (h->eaCoop) {
            if (launchK(h, K_EDGE_ANGLE, [&] {
                    hipLaunchKernelGGL(k_edge_angle_coop, dim3((m.nPoints + kEaPointsPerBlock - 1) / kEaPointsPerBlock), dim3(kBlock),
                                       sizeof(double) * 9 * (size_t)h->eaMaxEntries, h->stream, m, s, prm, h->eaMaxEntries, eaMaybe);
                })) return 1;
        } else if (launchK(h, K_EDGE_ANGLE, [&] { hipLaunchKernelGGL(k_edge_angle, dim3(gP), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
    }
    if (h->prm.faceAngleConstraint) {
        const uint8_t* faMaybe = nullptr;
        if (filt && h->faFilterInFlight) {
            if (depWait(h, DEP_JOIN, h->stream, h->evJoin)) return 1;   // join the side stream
            h->faFilterInFlight = false;
            faMaybe = h->dFaMaybe;
        } else if (filt) {
            h->faExactOnSide = false;
            if (nextFaGen(h, h->stream)) return 1;
            s.faGen = h->st.faGen;
            if (launchK(h, K_FA_FILTER, [&] {
                    if (h->edgeTilesOk)
                        hipLaunchKernelGGL(k_fa_filter_tile<256>, dim3(tileGrid(h->etl.nTiles, h->xcdMap)), dim3(256), h->edgeLds, h->stream, s, prm, h->ev, m.edges, h->dFaMaybe,
                                           h->etl.nTiles, h->xcdMap);
                    else
                        hipLaunchKernelGGL(k_fa_edges_filter, dim3(gridFor(m.nEdges)), dim3(kBlock), 0, h->stream, m, s, prm, h->dFaMaybe);
                })) return 1;
            faMaybe = h->dFaMaybe;
        } else {   // no filter: every point gets a fresh mark from k_fa_points
            h->faExactOnSide = false;
            if (nextFaGen(h, h->stream)) return 1;
            s.faGen = h->st.faGen;
        }
        if (h->faExactOnSide) h->faExactOnSide = false;          // done behind the filter on the side stream
        else if (runFaExactPass(h, s, faMaybe, h->stream)) return 1;
        if (h->walkMode < 0 && decideWalkMode(h)) return 1;
        if (h->walkMode == 1) {
            if (runHostWalk(h)) return 1;
        } else if (h->walkMode == 2) {
            if (runFixWalk(h)) return 1;
        } else {
            if (launchK(h, K_FA_PRED, [&] { hipLaunchKernelGGL(k_fa_pred, dim3(gP), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
            if (launchK(h, K_FA_WALK, [&] { hipLaunchKernelGGL(k_fa_walk, dim3(1), dim3(64), 0, h->stream, m, s); })) return 1;
        }
    }
    return 0;
}

static int checkDeviceError(smgpu_handle* h) {
    Accum a;
    HIP_OK(hipMemcpyAsync(&a, h->st.acc, sizeof(Accum), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    // reported once: the word is cleared, so that the handle stays usable (a peer-store wait that timed out makes every later
    // wait of the run return at once -- until the host has seen the error and, say, reconfigured the halo)
    if (a.err != 0) HIP_OK(hipMemsetAsync(&h->st.acc->err, 0, sizeof(int), h->stream));
    if (a.err == 3) return fail("face-angle walk: the workgroups of the device replay did not all become resident (grid barrier timed out); set SMGPU_WALK=host or lower SMGPU_WALK_BLOCKS");
    if (a.err == 1) return fail("Failed to find cLabel1/cLabel2: a point has fewer than two usable edge neighbours (SM.C:354-362)");
    if (a.err == 2) return fail("a shared point has more sharing ranks than supported");
    if (a.err == ROLE_ERR_TIMEOUT) return fail("a role of a multi-role launch (k_geom_halo / k_smooth_halo) waited two seconds for the workgroups of an earlier role; set SMGPU_HALO_MERGED=0");
    if (a.err == PUSH_ERR_TIMEOUT) return fail("peer-store transport: a peer's records did not arrive within the time limit (SMGPU_PUSH_TIMEOUT_S, default 60 s: a rank is gone, or not using the same transport)");
    if (a.err == BND_ERR_NORMAL) return fail("pointNormal is zero for a boundary point that is to be projected (BPS.C:691-696, OBB.C:609-610)");
    if (a.err == BND_ERR_NOHIT) return fail("Did not find surface intersection for a boundary point (BPS.C:932-938)");
    if (a.err == BND_ERR_STRING) return fail("Internal sanity check failed: Did not find any edges with the string index of a feature edge point (BPS.C:258-261)");
    return 0;
}

// the end-of-iteration reduction left to the next geometry launch, when no such launch is coming
static int flushDeferred(smgpu_handle* h) {
    if (h->deferN <= 0) return 0;
    HIP_OK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishBlock), 0, h->stream, h->st, h->deferN, h->deferIter, -1.0, h->deferLocal, h->deferHist);
    h->deferN = 0;
    h->deferLocal = h->deferHist = nullptr;
    HIP_OK(hipGetLastError());
    return 0;
}

int smgpu_iterate(smgpu_handle* h, int32_t nIters, double relTol, smgpu_iter_stats* stats, int32_t* nDone) {
    if (!h) return fail("null handle");
    if (!h->prmSet) return fail("smgpu_set_params has not been called");
    if (h->haloOn) return fail("smgpu_iterate is the single-rank loop; use smgpu_iter_begin/mid/end with a halo");
    if (nIters < 0) return fail("nIters < 0");
    if (nDone) *nDone = 0;
    if (nIters == 0) return 0;
    HIP_OK(hipSetDevice(h->device));
    if (h->statsCap < nIters) {
        if (h->dStats) {   // outgrown: release it (nothing is in flight between two smgpu_iterate calls)
            HIP_OK(hipStreamSynchronize(h->stream));
            h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), (void*)h->dStats), h->allocs.end());
            h->deviceBytes -= (int64_t)(sizeof(smgpu_iter_stats) * (size_t)h->statsCap);
            (void)hipFree(h->dStats);
            h->dStats = nullptr;
            h->statsCap = 0;
        }
        if (devAlloc(h, &h->dStats, (size_t)nIters)) return 1;
        h->statsCap = nIters;
    }
    HIP_OK(hipMemsetAsync(h->dStats, 0, sizeof(smgpu_iter_stats) * (size_t)nIters, h->stream));
    HIP_OK(hipMemsetAsync(h->st.acc, 0, sizeof(Accum), h->stream));
    h->st.stats = h->dStats;
    const MeshView& m = h->mv;
    const Prm prm = makePrm(h);
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    const int gP = gridFor(m.nPoints);
    // constraints on: the proposal array becomes the next coordinates (pointer swap) and k_apply_swap only restores the points
    // that do not move, instead of k_apply's copy of everything into a third array (SMGPU_APPLY_SWAP=0: that form).  Not with
    // boundary point smoothing: k_bnd_fix rewrites proposals behind the proposal kernel.
    const bool swapApply = swapApplyOn(h);
    const int gSwap = (int)((m.nPoints + (int64_t)kApplyPer * kBlock - 1) / ((int64_t)kApplyPer * kBlock));
    h->st.stepSqr = swapApply ? h->dStepSqr : nullptr;
    double*& other = swapApply ? h->st.prop : h->st.ptsNext;
    double* const buf0 = h->st.ptsCur;
    double* const buf1 = other;
    int launched = 0;
    // relTol <= 0 cannot stop the loop (residual >= 0): the end-of-iteration reduction then rides in the next
    // iteration's geometry launch instead of a launch of its own; the last iteration is closed by k_finish
    const bool deferFinish = relTol <= 0.0 && h->useTiles && h->geomT >= 64 && envInt("SMGPU_DEFER_FINISH", 1);
    if (flushDeferred(h)) return 1;
    for (int i = 0; i < nIters; ++i) {
        if (runBndPre(h)) return 1;
        if (runGeometry(h, nullptr, 0, false, true)) return 1;
        State s = h->st;
        if (fused) {
            if (runSmooth<true>(h, m, s, prm)) return 1;
        } else {
            if (runProposalAndConstraints(h)) return 1;
            if (swapApply) { if (launchK(h, K_APPLY, [&] { hipLaunchKernelGGL(k_apply_swap, dim3(gSwap), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1; }
            else if (launchK(h, K_APPLY, [&] { hipLaunchKernelGGL(k_apply, dim3(gP), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
        }
        const int nPart = ((fused && h->useTiles) ? h->stl.nTiles : (swapApply ? gSwap : gP)) + ((fused && h->bndOn) ? gridFor(2 * (int64_t)h->bv.nB) : 0);
        if (deferFinish && i + 1 < nIters) { h->deferN = nPart; h->deferIter = i; }
        else if (launchK(h, K_FINISH, [&] { hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishBlock), 0, h->stream, s, nPart, i, relTol, (double*)nullptr, (double*)nullptr); })) return 1;
        std::swap(h->st.ptsCur, other);  // mesh.movePoints, SM.C:2399
        ++launched;
        // a positive relTol can stop the loop: poll the device flag now and then so a converged run
        // does not queue thousands of no-op launches (relTol <= 0 can never stop: residual >= 0)
        if (relTol > 0.0 && (i % 8) == 7 && i + 1 < nIters) {
            int stop = 0;
            HIP_OK(hipMemcpyAsync(&stop, &h->st.acc->stop, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
            if (stop) break;
        }
    }
    std::vector<smgpu_iter_stats> hs((size_t)launched);
    HIP_OK(hipMemcpyAsync(hs.data(), h->dStats, sizeof(smgpu_iter_stats) * (size_t)launched, hipMemcpyDeviceToHost, h->stream));
    if (checkDeviceError(h)) return 1;
    int done = 0;
    while (done < launched && (hs[done].nNearTies & kStatsWritten)) ++done;
    for (int i = 0; i < done; ++i) hs[i].nNearTies &= kStatsWritten - 1;
    if (stats) std::memcpy(stats, hs.data(), sizeof(smgpu_iter_stats) * (size_t)done);
    if (nDone) *nDone = done;
    // the coordinates of iteration `done` live in buf1 when done is odd, buf0 when even
    h->st.ptsCur = (done & 1) ? buf1 : buf0;
    other = (done & 1) ? buf0 : buf1;
    h->st.stepSqr = nullptr;
    h->st.stats = nullptr;
    if (drainTimers(h)) return 1;
    if (h->dWalkMemo) {      // per smgpu_iterate call: stars whose inputs repeat the point's previous walk bit for bit
        unsigned long long v[2];
        HIP_OK(hipMemcpy(v, h->dWalkMemo, sizeof(v), hipMemcpyDeviceToHost));
        const unsigned long long dm = v[0] - h->walkMemoLast[0], dt = v[1] - h->walkMemoLast[1];
        if (dt > 0) std::fprintf(stderr, "[smgpu] walk memo: %llu of %llu stars (%.1f %%) had the inputs of the point's previous walk, over %d iterations\n", dm, dt, 100.0 * (double)dm / (double)dt, done);
        h->walkMemoLast[0] = v[0]; h->walkMemoLast[1] = v[1];
    }
    if (h->fixAlloc && envInt("SMGPU_WALK_STATS", 0)) {
        int v[16];
        HIP_OK(hipMemcpy(v, h->fxw.flags, sizeof(v), hipMemcpyDeviceToHost));
        if (v[12] > 0) std::fprintf(stderr, "[smgpu] walk replay: %d launches, per launch %.1f outer rounds, %.1f sweep votes, %.1f us in the kernel's loop, %.1f us of it in sweeps\n",
                                    v[12], (double)v[8] / v[12], (double)v[9] / v[12], 0.01 * v[10] / v[12], 0.01 * v[11] / v[12]);
    }
    return 0;
}

int smgpu_set_device_share(smgpu_handle* h, int32_t nEngines) {
    if (!h || nEngines < 1) return fail("smgpu_set_device_share: bad argument");
    h->deviceShare = nEngines;
    if (h->fixAlloc) h->walkFixBlocks = std::max(1, std::min(envInt("SMGPU_WALK_BLOCKS", std::max(8, 128 / nEngines)), 256));
    return 0;
}

int smgpu_debug_walk_mode(smgpu_handle* h, int32_t* mode, int32_t* switches, int32_t* lastCount) {
    if (!h || !mode || !switches || !lastCount) return fail("null argument");
    *mode = h->walkMode;
    *switches = h->walkSwitches;
    *lastCount = h->nActiveHost ? *(volatile int*)h->nActiveHost : -1;
    return 0;
}

int smgpu_debug_halo_mode(smgpu_handle* h, int32_t* multiRole, int32_t* flaggedOut, int32_t* fixInside) {
    if (!h || !multiRole || !flaggedOut || !fixInside) return fail("null argument");
    *multiRole = h->mergedIter ? 1 : 0;
    *flaggedOut = (h->mergedIter && h->useExch) ? 1 : 0;
    *fixInside = (h->mergedIter && h->fixInSmooth) ? 1 : 0;
    return 0;
}

int smgpu_get_points(smgpu_handle* h, double* out) {
    if (!h || !out) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipMemcpyAsync(out, h->st.ptsCur, sizeof(double) * 3 * (size_t)h->mv.nPoints, hipMemcpyDeviceToHost, h->stream));
    // the step-wise loop (smgpu_iter_begin / mid / end) never synchronises on its own: an error word raised by a kernel (a grid
    // barrier or peer-store wait that timed out, a point without usable neighbours ...) surfaces here at the latest, with the
    // coordinates it spoiled
    return checkDeviceError(h);
}

int smgpu_get_near_ties(smgpu_handle* h, int64_t out[4]) {
    if (!h || !out) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    unsigned long long v[4] = {0, 0, 0, 0};
    HIP_OK(hipMemcpyAsync(v, h->st.nearTotal, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    out[1] = (int64_t)v[0]; out[2] = (int64_t)v[1]; out[3] = (int64_t)v[2];
    out[0] = out[1] + out[2] + out[3];
    return 0;
}

int smgpu_check_error(smgpu_handle* h) {
    if (!h) return fail("null handle");
    HIP_OK(hipSetDevice(h->device));
    return checkDeviceError(h);
}

int smgpu_set_points(smgpu_handle* h, const double* pts) {
    if (!h || !pts) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    h->geomAheadDone = false;
    HIP_OK(hipMemcpyAsync(h->st.ptsCur, pts, sizeof(double) * 3 * (size_t)h->mv.nPoints, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}

int smgpu_enable_timing(smgpu_handle* h, int32_t on) {
    if (!h) return fail("null handle");
    if (!on && drainTimers(h)) return 1;
    h->timing = on != 0;
    return 0;
}

int smgpu_get_counters(smgpu_handle* h, smgpu_counters* o) {
    if (!h || !o) return fail("null argument");
    if (drainTimers(h)) return 1;
    if (h->dWalkOps && h->launches[K_FA_PRED] > 0) {   // the walk predicates' algorithmic FP64 instructions, counted by the timed launches
        unsigned long long ops[64];
        HIP_OK(hipMemcpy(ops, h->dWalkOps, sizeof(ops), hipMemcpyDeviceToHost));
        unsigned long long tot = 0;
        for (unsigned long long v : ops) tot += v;
        if (tot > 0) { h->algoF64[K_FA_PRED] = (int64_t)(tot / (unsigned long long)h->launches[K_FA_PRED]); h->algoBytes[K_FA_PRED] = 0; }
    }
    if (h->walkMode > 0) h->algoBytes[K_FA_WALK] = 0;   // the compacted replay is latency work: no streaming byte count to price it against
    o->nKernels = K_COUNT;
    for (int k = 0; k < K_COUNT; ++k) {
        o->name[k] = kKernelNames[k];
        o->ms[k] = h->ms[k];
        o->launches[k] = h->launches[k];
        o->algoBytesPerLaunch[k] = h->algoBytes[k];
        o->algoF64OpsPerLaunch[k] = h->algoF64[k];
    }
    return 0;
}

int smgpu_reset_counters(smgpu_handle* h) {
    if (!h) return fail("null handle");
    if (drainTimers(h)) return 1;
    if (h->dWalkOps) HIP_OK(zeroNow(h->dWalkOps, 0, 64 * sizeof(unsigned long long)));
    for (int k = 0; k < K_COUNT; ++k) { h->ms[k] = 0; h->launches[k] = 0; }
    return 0;
}

// ---- multi-rank ----------------------------------------------------------------------------------
static int exchAfterCompute(smgpu_handle* h);
int smgpu_halo_set_stats_history(smgpu_handle* h, void* history, int32_t capacity) {
    if (!h) return fail("null handle");
    if (history && capacity <= 0) return fail("smgpu_halo_set_stats_history: capacity must be positive");
    // closes the last iteration of a loop; the host reads the history on its exchange stream next
    const bool pendingRecord = h->deferN > 0;
    if (flushDeferred(h)) return 1;
    if (pendingRecord && h->haloOn && exchAfterCompute(h)) return 1;
    h->statsHistory = (double*)history;
    h->statsHistoryCap = history ? capacity : 0;
    h->statsHistoryN = 0;
    return 0;
}

int smgpu_halo_l_doubles(smgpu_handle* h, int32_t* doublesPerSlot) {
    if (!h || !doublesPerSlot) return fail("null argument");
    *doublesPerSlot = h->st.lStride > 0 ? h->st.lStride : SMGPU_HALO_L_LAYERS;
    return 0;
}

int smgpu_get_stream(smgpu_handle* h, void** stream) {
    if (!h || !stream) return fail("null argument");
    *stream = (void*)h->stream;
    return 0;
}

int smgpu_halo_set_exchange_stream(smgpu_handle* h, int32_t useExchangeStream, void* exchangeStream) {
    if (!h) return fail("null handle");
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipDeviceSynchronize());
    if (depInit(h)) return 1;
    if (h->pushOn && useExchangeStream && (hipStream_t)exchangeStream != h->stream)
        return fail("smgpu_halo_set_exchange_stream: the peer-store transport is on (nothing is enqueued by the host: no exchange stream)");
    h->useExch = useExchangeStream && (hipStream_t)exchangeStream != h->stream;
    h->exch = (hipStream_t)exchangeStream;
    if (h->useExch && !h->evToExch) {
        if (hipEventCreateWithFlags(&h->evToExch, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->evFromExch, hipEventDisableTiming) != hipSuccess)
            return fail("event creation failed");
    }
    return 0;
}

int smgpu_halo_configure(smgpu_handle* h, const smgpu_halo_desc* d) {
    if (!h || !d) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    // the host's buffers may still be being initialised on the host's streams
    HIP_OK(hipDeviceSynchronize());
    // a new slot layout invalidates the peer-store transport's destination tables and restarts the flag tags: the caller describes
    // the peers again (smgpu_halo_set_push) after every configure
    h->pushOn = false; h->st.push = PushView{};
    h->nShared = d->nShared; h->nSend = d->nSend; h->nRecv = d->nRecv;
    if (smgpu_halo_set_exchange_stream(h, d->useExchangeStream, d->exchangeStream)) return 1;
    const int P = h->mv.nPoints;
    std::vector<int> sharedLocal(d->sharedLocal, d->sharedLocal + d->nShared);
    std::vector<int> sendShared(d->sendShared, d->sendShared + d->nSend);
    std::vector<int> combOff(d->combOffsets, d->combOffsets + d->nShared + 1);
    std::vector<int> combSlots(d->combSlots, d->combSlots + combOff[d->nShared]);
    std::vector<int> slot((size_t)P, -1);
    for (int i = 0; i < d->nShared; ++i) {
        if (sharedLocal[i] < 0 || sharedLocal[i] >= P) return fail("halo: shared point id out of range");
        slot[sharedLocal[i]] = i;
        if (combOff[i + 1] - combOff[i] > kMaxSharers) return fail("halo: more than 16 ranks share a point");
    }
    for (int v : sendShared) if (v < 0 || v >= d->nShared) return fail("halo: sendShared out of range");
    // send slots of each shared point (the pack kernels write the own record and its copies in one pass)
    std::vector<int> sendOff((size_t)d->nShared + 1, 0), sendSlots((size_t)d->nSend + 1, 0);
    for (int v : sendShared) ++sendOff[(size_t)v + 1];
    for (int i = 0; i < d->nShared; ++i) sendOff[(size_t)i + 1] += sendOff[(size_t)i];
    {
        std::vector<int> fill(sendOff.begin(), sendOff.end() - 1);
        for (int k = 0; k < d->nSend; ++k) sendSlots[(size_t)fill[(size_t)sendShared[(size_t)k]]++] = k;
    }
    const int *so = nullptr, *ss = nullptr;
    if (devUpload(h, &so, sendOff) || devUpload(h, &ss, sendSlots)) return 1;
    h->dSendOff = (int*)so; h->dSendSlots = (int*)ss;
    for (int v : combSlots) if (v < -1 || v >= d->nRecv) return fail("halo: combSlots out of range");
    const int *a = nullptr, *b = nullptr, *c = nullptr, *e = nullptr, *f = nullptr;
    if (devUpload(h, &a, sharedLocal) || devUpload(h, &b, sendShared) || devUpload(h, &c, combOff) ||
        devUpload(h, &e, combSlots) || devUpload(h, &f, slot)) return 1;
    h->dSharedLocal = (int*)a; h->dSendShared = (int*)b; h->dCombOff = (int*)c; h->dCombSlots = (int*)e; h->dSharedSlot = (int*)f;
    if (devAlloc(h, &h->dOwnA, (size_t)d->nShared * SMGPU_HALO_A_DOUBLES)) return 1;
    if (devAlloc(h, &h->dCombA, (size_t)d->nShared * SMGPU_HALO_A_DOUBLES)) return 1;
    h->sendA = (double*)d->sendA; h->recvA = (double*)d->recvA;
    h->sendL = (double*)d->sendL; h->recvL = (double*)d->recvL;
    h->sharedLocalHost = sharedLocal;
    {
        std::vector<int> multi;
        bool tooMany = false;
        for (int i = 0; i < d->nShared; ++i) {
            const int n = combOff[(size_t)i + 1] - combOff[(size_t)i];
            if (n > 16) tooMany = true;
            else if (n > 2) multi.push_back(i);
        }
        h->nMulti = 0; h->dMultiIdx = nullptr;
        if (!tooMany) {      // otherwise the one-lane form handles every point (and reports > kMaxSharers)
            std::vector<int> mslots(multi.size() * 16 + 16, -2);
            for (size_t g = 0; g < multi.size(); ++g) {
                const int i = multi[g];
                for (int j = combOff[(size_t)i]; j < combOff[(size_t)i + 1]; ++j) mslots[g * 16 + (size_t)(j - combOff[(size_t)i])] = combSlots[(size_t)j];
            }
            multi.push_back(0);   // never an empty upload; the extra entry is not counted
            const int *pm = nullptr, *ps = nullptr;
            if (devUpload(h, &pm, multi) || devUpload(h, &ps, mslots)) return 1;
            h->dMultiIdx = (int*)pm;
            h->dMultiSlots = (int*)ps;
            h->nMulti = (int)multi.size() - 1;
            // two-sharer points: the other rank's receive slot (k_halo_combineA2)
            std::vector<int> peer((size_t)d->nShared + 1, -1);
            for (int i = 0; i < d->nShared; ++i) {
                const int b = combOff[(size_t)i];
                if (combOff[(size_t)i + 1] - b != 2) continue;
                const int s0 = combSlots[(size_t)b], s1 = combSlots[(size_t)b + 1];
                const int other = s0 < 0 ? s1 : s0;
                if (other >= 0 && other < 0x40000000) peer[(size_t)i] = other | (s0 < 0 ? 0x40000000 : 0);
                else { h->dMultiIdx = nullptr; break; }      // (cannot happen: a two-sharer point has exactly one own entry)
            }
            const int* pp = nullptr;
            if (devUpload(h, &pp, peer)) return 1;
            h->dPeer = (int*)pp;
        }
    }
    if (h->layersOn) return fail("smgpu_halo_configure after the boundary layer set-up: configure the halo first");
    if (h->bndOn) return fail("smgpu_halo_configure: boundary point smoothing is a serial-run feature");
    h->sendF = (int*)d->sendF; h->recvF = (int*)d->recvF;
    h->localStats = (double*)d->localStats;
    if ((d->nSend && (!h->sendA || !h->sendF)) || (d->nRecv && (!h->recvA || !h->recvF)) || !h->localStats)
        return fail("halo: null exchange buffer");
    if (h->useTiles) {
        std::vector<int> inter, shr;
        for (int ti = 0; ti < h->stl.nTiles; ++ti) {
            bool any = false;
            for (int pi = h->stl.ptBeg[ti]; pi < h->stl.ptBeg[ti + 1] && !any; ++pi) any = slot[(size_t)h->stl.order[(size_t)pi]] >= 0;
            (any ? shr : inter).push_back(ti);
        }
        const int *pa = nullptr, *pb = nullptr;
        if (devUpload(h, &pa, inter) || devUpload(h, &pb, shr)) return 1;
        h->dInteriorTiles = (int*)pa; h->dSharedTiles = (int*)pb;
        h->nInteriorTiles = (int)inter.size(); h->nSharedTiles = (int)shr.size();
        // geometry tiles: a tile is "shared" when one of its cells has a shared point
        const Topology& t = h->topo;
        std::vector<int> gi, gs;
        for (int ti = 0; ti < h->gt.nTiles; ++ti) {
            bool any = false;
            for (int ci = h->gt.cellBeg[ti]; ci < h->gt.cellBeg[ti + 1] && !any; ++ci) {
                const int c = h->gt.order[(size_t)ci];
                for (int k = t.cellFacesGeom.off[c]; k < t.cellFacesGeom.off[c + 1] && !any; ++k) {
                    const int f = t.cellFacesGeom.val[k] & 0x7fffffff;
                    for (int j = t.facePoints.off[f]; j < t.facePoints.off[f + 1] && !any; ++j) any = slot[(size_t)t.facePoints.val[j]] >= 0;
                }
            }
            (any ? gs : gi).push_back(ti);
        }
        const int *pc = nullptr, *pd = nullptr;
        if (devUpload(h, &pc, gi) || devUpload(h, &pd, gs)) return 1;
        h->dGeomInterior = (int*)pc; h->dGeomShared = (int*)pd;
        h->nGeomInterior = (int)gi.size(); h->nGeomShared = (int)gs.size();
        // per tile position of the regular tiles: the point's shared slot (k_smooth_halo's regular tiles skip their shared points)
        std::vector<int> posSlot((size_t)P, -1);
        for (int pi = 0; pi < P; ++pi) posSlot[(size_t)pi] = slot[(size_t)h->stl.order[(size_t)pi]];
        const int* pe = nullptr;
        if (devUpload(h, &pe, posSlot)) return 1;
        h->dPosSlot = (int*)pe;
        // tiles over the shared points only, in the regular tiles' (Morton) order: the halo roles of k_geom_halo / k_smooth_halo
        h->shr = SmoothTiles();
        h->hv = SmoothTileView{};
        h->haloLds = 0;
        if (d->nShared > 0 && (int)h->isInternalHost.size() == P) {
            std::vector<int32_t> sub;
            sub.reserve((size_t)d->nShared);
            for (int pi = 0; pi < P; ++pi) if (posSlot[(size_t)pi] >= 0) sub.push_back(h->stl.order[(size_t)pi]);
            std::vector<double> pts(3 * (size_t)P);      // (the builder only reads coordinates when it orders the points itself)
            const int capSC = envInt("SMGPU_SMOOTH_CAPC", std::min(2 * h->smoothT, 1500)), capSN = envInt("SMGPU_SMOOTH_CAPN", std::min(3 * h->smoothT, 1500));
            std::string err = h->shr.buildBoundaries(h->topo, pts.data(), false, h->smoothT, capSC, capSN, nullptr, &sub, envInt("SMGPU_SMOOTH_CAPTOTAL", defaultSmoothCapTotal(h->smoothT)));
            if (!err.empty()) return fail("shared-point tiles: " + err);
            // the tables on the device where the addressing lives there (tiles_dev.hip; the corner lists they read stayed there),
            // else on the host -- which first fetches those lists
            int onDev = 1;
            SmoothTilesDev sd;
            if (h->devLists.valid && envInt("SMGPU_DEVICE_TILES", 1) == 1) {
                std::string why;
                onDev = buildSmoothTablesOnDevice(h->shr, h->devLists, P, h->topo.maxPointPoints, h->isInternalHost.data(), h->device, sd, why);
                if (onDev == 2) return fail("shared-point tiles: " + why);
            }
            if (onDev == 0) {
                SmoothTileView& v = h->hv;
                auto adoptS = [&](auto*& dst, const SmoothTilesDev::Arr& a) { dst = (std::remove_reference_t<decltype(dst)>)a.p; h->allocs.push_back(a.p); h->deviceBytes += (int64_t)a.bytes; };
                adoptS(v.ptOrder, sd.order); adoptS(v.ptBeg, sd.ptBeg); adoptS(v.tcIds, sd.tcIds); adoptS(v.tnIds, sd.tnIds); adoptS(v.selfLoc, sd.selfLoc);
                adoptS(v.pcEll, sd.pcEll); adoptS(v.ppEll, sd.ppEll); adoptS(v.pairEll, sd.pairEll); adoptS(v.pfEll, sd.pfEll); adoptS(v.meta, sd.meta);
                int rcU = 0;
                rcU |= devUpload(h, &v.tcOff, h->shr.tcOff); rcU |= devUpload(h, &v.tnOff, h->shr.tnOff);
                rcU |= devUpload(h, &v.pcBase, h->shr.pcBase); rcU |= devUpload(h, &v.pcWidth, h->shr.pcWidth);
                rcU |= devUpload(h, &v.ppBase, h->shr.ppBase); rcU |= devUpload(h, &v.ppWidth, h->shr.ppWidth);
                rcU |= devUpload(h, &v.pfBase, h->shr.pfBase); rcU |= devUpload(h, &v.pfWidth, h->shr.pfWidth);
                if (rcU) return 1;
                v.maxCells = h->shr.maxCells; v.maxPoints = h->shr.maxPoints;
                v.usePairShare = h->sv.usePairShare;
            } else {
                if (ensureHostLists(h, 1)) return 1;      // (the tables' corner lists: pointFaces with prev / next)
                err = h->shr.buildTables(h->topo, h->isInternalHost.data(), true);
                if (!err.empty()) return fail("shared-point tiles: " + err);
                if (uploadSmoothView(h, h->shr, h->hv, h->sv.usePairShare)) return 1;
            }
            if (envInt("SMGPU_VERBOSE", 0) >= 2) std::fprintf(stderr, "[smgpu] shared-point tiles: %d tiles, tables on the %s\n", h->shr.nTiles, onDev == 0 ? "device" : "host");
            h->haloLds = sizeof(double) * 3 * maxTileTotal(h->shr.nTiles, {&h->shr.tcOff, &h->shr.tnOff});
            // per position of those tiles: slot, two-sharer peer code (k_halo_combineA2's table), send slots
            const size_t nS = sub.size();
            std::vector<int> spSlot(nS), spPeer(nS, -1), spDst0(nS, -1), spNDst(nS, 0);
            for (size_t i = 0; i < nS; ++i) {
                const int sl = slot[(size_t)sub[i]];
                spSlot[i] = sl;
                const int b = combOff[(size_t)sl];
                if (combOff[(size_t)sl + 1] - b == 2) {
                    const int s0 = combSlots[(size_t)b], s1 = combSlots[(size_t)b + 1];
                    const int other = s0 < 0 ? s1 : s0;
                    if (other >= 0 && other < 0x40000000) spPeer[i] = other | (s0 < 0 ? 0x40000000 : 0);
                }
                spNDst[i] = sendOff[(size_t)sl + 1] - sendOff[(size_t)sl];
                if (spNDst[i] > 0) spDst0[i] = sendSlots[(size_t)sendOff[(size_t)sl]];
            }
            const int *qa = nullptr, *qb = nullptr, *qc = nullptr, *qd = nullptr;
            if (devUpload(h, &qa, spSlot) || devUpload(h, &qb, spPeer) || devUpload(h, &qc, spDst0) || devUpload(h, &qd, spNDst)) return 1;
            h->st.spSlot = qa; h->st.spPeer = qb; h->st.spDst0 = qc; h->st.spNDst = qd;
        }
        if (!h->dRoleTickets) { if (devAlloc(h, &h->dRoleTickets, 3 * (size_t)kRoleWords)) return 1; }
        HIP_OK(zeroNow(h->dRoleTickets, 0, 3 * (size_t)kRoleWords * sizeof(unsigned)));
        h->roleLaunches[0] = h->roleLaunches[1] = h->roleLaunches[2] = 0;
        if (devAlloc(h, &h->dOwnF, (size_t)std::max(d->nShared, 1))) return 1;
        h->mergedWanted = envInt("SMGPU_HALO_MERGED", 1) != 0;
        h->flagWanted = envInt("SMGPU_HALO_FLAGGED", 1) != 0;
        h->flagBuilt = false;
    }
    {   // partial slots: tiles (or point blocks) + the blocks of k_shared_fix
        const size_t nPart = (size_t)std::max(gridFor(P), h->useTiles ? h->stl.nTiles : 0) + (size_t)gridFor(d->nShared) + 2 * (size_t)gridFor(P) + 2;
        if (devAlloc(h, &h->st.blkMax, nPart) || devAlloc(h, &h->st.blkCnt, nPart)) return 1;
    }
    h->st.sharedSlot = h->dSharedSlot;
    h->st.posSlot = h->dPosSlot;
    h->st.combA = h->dCombA;
    h->st.combOff = h->dCombOff; h->st.combSlots = h->dCombSlots; h->st.ownA = h->dOwnA; h->st.recvA = h->recvA;
    h->st.sendOff = h->dSendOff; h->st.sendSlots = h->dSendSlots; h->st.sendF = h->sendF;
    // two-sharer points combined inside the smoothing kernel (needs the tiled kernels and the table of the multi-sharer points).
    // MEASURED (one rank of eight, 30 k shared points): slower, 211 against 199 us per iteration -- the combine is a chain of
    // dependent loads (offset -> slot -> record) and stalls whole waves of the smoothing kernel (47 -> 59 us), which costs more
    // than the two small launches it removes.  Off by default (SMGPU_HALO_INLINE=1 selects it; results are the same).
    h->st.inlineCombine = 0;     // (the knob SMGPU_HALO_INLINE is gone, see k_smooth_tile)
    h->st.inlinePackF = (h->useTiles && envInt("SMGPU_HALO_INLINE_PACKF", 0)) ? 1 : 0;
    h->packTiles = envInt("SMGPU_PACK_TILES", 1) != 0;
    if (h->useTiles) {
        ensureDynLds(k_pack_tile<64>, h->device, h->smoothLds);
        ensureDynLds(k_pack_tile<128>, h->device, h->smoothLds);
        ensureDynLds(k_pack_tile<256>, h->device, h->smoothLds);
        ensureDynLds(k_smooth_halo<256>, h->device, std::max(h->smoothLds, h->haloLds));
        ensureDynLds(k_geom_halo<256, false>, h->device, std::max(h->geomLds, h->haloLds));
        ensureDynLds(k_geom_halo<256, true>, h->device, std::max(h->geomLds, h->haloLds));
    }
    h->st.lStride = SMGPU_HALO_L_LAYERS;
    h->haloOn = true;
    h->haloIter = 0;
    HIP_OK(hipMemsetAsync(h->st.acc, 0, sizeof(Accum), h->stream));
    return 0;
}

// order the host's exchange stream after everything queued on the engine's stream so far / the engine's stream
// after everything the host has queued on its exchange stream so far
static int exchAfterCompute(smgpu_handle* h) {
    if (!h->useExch) return 0;
    if (depSignal(h, DEP_TO_EXCH, h->stream, h->evToExch) || depWait(h, DEP_TO_EXCH, h->exch, h->evToExch)) return 1;
    return 0;
}
static int computeAfterExch(smgpu_handle* h) {
    if (!h->useExch) return 0;
    if (depSignal(h, DEP_FROM_EXCH, h->exch, h->evFromExch) || depWait(h, DEP_FROM_EXCH, h->stream, h->evFromExch)) return 1;
    return 0;
}

// what the first kernel that consumes exchange `kind` (0 = A + L, 1 = F) of this iteration waits for
static PushWait pushWaitOf(const smgpu_handle* h, int kind) {
    PushWait pw;
    if (h->mergedIter && h->useExch) {      // the flagged arrangement: the word the exchange stream writes behind the host's exchange
        pw.localFlag = h->flagView.localFlag; pw.nPeers = 1; pw.kind = kind; pw.tag = (unsigned)(h->haloIter + 1); pw.err = &h->st.acc->err; pw.fence = 0;
        pw.timeoutTicks = (unsigned long long)std::max(1, envInt("SMGPU_PUSH_TIMEOUT_S", 60)) * 100000000ull;
        return pw;
    }
    pw.localFlag = h->pushOn ? h->st.push.localFlag : nullptr;
    pw.nPeers = h->pushPeers; pw.kind = kind; pw.tag = (unsigned)(h->haloIter + 1); pw.err = &h->st.acc->err; pw.fence = h->st.push.fence;
    pw.timeoutTicks = h->pushTimeoutTicks;
    return pw;
}

// (re)build the per-slot destination tables of the peer-store transport; the L records' stride follows smgpu_halo_l_doubles
static int pushBuildTables(smgpu_handle* h) {
    const int w = h->st.lStride > 0 ? h->st.lStride : SMGPU_HALO_L_LAYERS;
    std::vector<void*> a((size_t)std::max(h->nSend, 1), nullptr), l(a.size(), nullptr), f(a.size(), nullptr), flags((size_t)std::max(h->pushPeers, 1), nullptr);
    int k = 0;
    for (int o = 0; o < h->pushPeers; ++o) {
        for (int j = 0; j < h->pushCount[(size_t)o]; ++j, ++k) {
            const size_t rs = (size_t)h->pushRemoteBase[(size_t)o] + (size_t)j;
            a[(size_t)k] = (double*)h->pushRecvA[(size_t)o] + rs * SMGPU_HALO_A_DOUBLES;
            l[(size_t)k] = h->pushRecvL[(size_t)o] ? (void*)((double*)h->pushRecvL[(size_t)o] + rs * (size_t)w) : nullptr;
            f[(size_t)k] = (int32_t*)h->pushRecvF[(size_t)o] + rs;
        }
        flags[(size_t)o] = (unsigned*)h->pushFlags[(size_t)o] + 2 * (size_t)h->pushMyIndex[(size_t)o];
    }
    if (k != h->nSend) return fail("smgpu_halo_set_push: the peers' slot counts do not add up to nSend");
    HIP_OK(hipMemcpy(h->dSlotA, a.data(), a.size() * sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->dSlotL, l.data(), l.size() * sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->dSlotF, f.data(), f.size() * sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->dPeerFlag, flags.data(), flags.size() * sizeof(void*), hipMemcpyHostToDevice));
    h->pushStride = w;
    return 0;
}

int smgpu_halo_set_push(smgpu_handle* h, const smgpu_push_desc* d) {
    if (!h || !h->haloOn) return fail("halo not configured");
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipDeviceSynchronize());
    if (!d) {
        h->pushOn = false; h->st.push = PushView{};
        h->st.inlineCombine = 0;
        h->st.inlinePackF = (h->useTiles && envInt("SMGPU_HALO_INLINE_PACKF", 0)) ? 1 : 0;      // as smgpu_halo_configure chose it
        return 0;
    }
    if (d->nPeers < 0 || d->nPeers > 64) return fail("smgpu_halo_set_push: at most 64 peers");
    if (h->nShared && !(h->useTiles && h->nSharedTiles > 0 && h->packTiles)) return fail("smgpu_halo_set_push: needs the tiled pack kernel (SMGPU_TILES / SMGPU_PACK_TILES)");
    if (h->useExch) return fail("smgpu_halo_set_push: the exchange stream arrangement does not apply (nothing is enqueued by the host)");
    h->pushPeers = d->nPeers;
    h->pushCount.assign(d->peerCount, d->peerCount + d->nPeers);
    h->pushRemoteBase.assign(d->remoteBase, d->remoteBase + d->nPeers);
    h->pushMyIndex.assign(d->myIndexAtPeer, d->myIndexAtPeer + d->nPeers);
    h->pushRecvA.assign(d->peerRecvA, d->peerRecvA + d->nPeers);
    h->pushRecvL.assign(d->peerRecvL, d->peerRecvL + d->nPeers);
    h->pushRecvF.assign(d->peerRecvF, d->peerRecvF + d->nPeers);
    h->pushFlags.assign(d->peerFlags, d->peerFlags + d->nPeers);
    h->pushLocalFlags = d->localFlags;
    for (int o = 0; o < d->nPeers; ++o)
        if (!h->pushRecvA[(size_t)o] || !h->pushRecvF[(size_t)o] || !h->pushFlags[(size_t)o] || h->pushMyIndex[(size_t)o] < 0 || h->pushMyIndex[(size_t)o] >= 64)
            return fail("smgpu_halo_set_push: incomplete peer description");
    if (!h->dSlotA) {
        const size_t n = (size_t)std::max(h->nSend, 1);
        void **a = nullptr, **l = nullptr, **f = nullptr, **pf = nullptr;
        if (devAlloc(h, &a, n) || devAlloc(h, &l, n) || devAlloc(h, &f, n) || devAlloc(h, &pf, 64) || devAlloc(h, &h->dPushTicket, 2)) return 1;
        h->dSlotA = a; h->dSlotL = l; h->dSlotF = f; h->dPeerFlag = pf;
        HIP_OK(zeroNow(h->dPushTicket, 0, 2 * sizeof(unsigned)));
    }
    if (pushBuildTables(h)) return 1;
    PushView pv;
    pv.slotA = (double* const*)h->dSlotA; pv.slotL = (double* const*)h->dSlotL; pv.slotF = (int* const*)h->dSlotF;
    pv.ticket = h->dPushTicket; pv.peerFlag = (unsigned* const*)h->dPeerFlag; pv.localFlag = (const unsigned*)h->pushLocalFlags; pv.nPeers = d->nPeers;
    pv.fence = envInt("SMGPU_PUSH_FENCE", 0) ? 1 : 0;
    h->pushTimeoutTicks = (unsigned long long)std::max(1, envInt("SMGPU_PUSH_TIMEOUT_S", 60)) * 100000000ull;   // s_memrealtime: 100 MHz
    h->st.push = pv;
    h->st.inlineCombine = 0;     // the records are consumed by the combine kernel (it carries the wait) ...
    h->st.inlinePackF = 0;       // ... and the flags leave through k_halo_packF (it carries the signal)
    h->pushOn = true;
    return 0;
}

// device memory another process can map (the receive buffers and the flag words of the peer-store transport): uncached, so that
// what a peer stores is what the next load returns -- no stale line in this GPU's L2
int smgpu_push_alloc(int32_t device, size_t bytes, void** ptr, void* ipcHandle64) {
    if (!ptr || !ipcHandle64 || bytes == 0) return fail("smgpu_push_alloc: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    HIP_OK(hipSetDevice(device));
    void* p = nullptr;
    HIP_OK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    HIP_OK(hipMemset(p, 0, bytes));
    HIP_OK(hipDeviceSynchronize());
    hipIpcMemHandle_t hd;
    const hipError_t e = hipIpcGetMemHandle(&hd, p);
    if (e != hipSuccess) { (void)hipFree(p); return fail(std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e)); }
    std::memcpy(ipcHandle64, &hd, 64);
    *ptr = p;
    return 0;
}
int smgpu_push_open(int32_t device, const void* ipcHandle64, void** ptr) {
    if (!ptr || !ipcHandle64) return fail("smgpu_push_open: bad argument");
    HIP_OK(hipSetDevice(device));
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, ipcHandle64, 64);
    HIP_OK(hipIpcOpenMemHandle(ptr, hd, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int smgpu_push_close(void* ptr) { if (ptr) HIP_OK(hipIpcCloseMemHandle(ptr)); return 0; }
int smgpu_push_free(void* ptr) { if (ptr) HIP_OK(hipFree(ptr)); return 0; }

// Constraints off, tiled kernels, exchanges in order (or peer stores): the iteration as two multi-role launches + k_shared_fix
// (kernels_tiled.hpp: k_geom_halo, k_smooth_halo) instead of six kernels.
static bool flagUsable(const smgpu_handle* h) {
    const char* v = std::getenv("ROCPROF_COUNTER_COLLECTION");      // (counter collection serialises kernels: a role would spin for a kernel that cannot start)
    return h->useExch && h->flagWanted && h->streamOps && !h->pushOn && h->deviceShare == 1 && !(v && std::atoi(v) != 0) && h->sendA && h->sendF;
}
// the PushView onto this rank's own send buffers and flag words (built once per halo configuration)
static int ensureFlagView(smgpu_handle* h) {
    if (h->flagBuilt) return 0;
    const size_t n = (size_t)std::max(h->nSend, 1);
    if (!h->dFlagWords) {
        void **a = nullptr, **f = nullptr, **pf = nullptr;
        if (devAlloc(h, &h->dFlagWords, 64) || devAlloc(h, &a, n) || devAlloc(h, &f, n) || devAlloc(h, &pf, 1) || devAlloc(h, &h->dSelfTicket, 2)) return 1;
        h->dSelfSlotA = a; h->dSelfSlotF = f; h->dSelfPeerFlag = pf;
    }
    std::vector<void*> a(n, nullptr), f(n, nullptr);
    for (int k = 0; k < h->nSend; ++k) { a[(size_t)k] = h->sendA + (size_t)k * SMGPU_HALO_A_DOUBLES; f[(size_t)k] = h->sendF + k; }
    void* w = h->dFlagWords;
    HIP_OK(hipMemcpy(h->dSelfSlotA, a.data(), n * sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->dSelfSlotF, f.data(), n * sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->dSelfPeerFlag, &w, sizeof(void*), hipMemcpyHostToDevice));
    HIP_OK(zeroNow(h->dFlagWords, 0, 64 * sizeof(uint32_t)));
    HIP_OK(zeroNow(h->dSelfTicket, 0, 2 * sizeof(unsigned)));
    PushView pv{};
    pv.slotA = (double* const*)h->dSelfSlotA; pv.slotL = nullptr; pv.slotF = (int* const*)h->dSelfSlotF;
    pv.ticket = h->dSelfTicket; pv.peerFlag = (unsigned* const*)h->dSelfPeerFlag; pv.localFlag = (const unsigned*)(h->dFlagWords + 16); pv.nPeers = 1;
    pv.fence = 0;
    h->flagView = pv;
    h->flagBuilt = true;
    return 0;
}
// one wave on the exchange stream: raise `writeWord` (behind what the host has enqueued there) and / or wait for `waitWord`
static int flagRelay(smgpu_handle* h, int writeIdx, int waitIdx) {
    const unsigned tag = (unsigned)(h->haloIter + 1);
    const unsigned long long ticks = (unsigned long long)std::max(1, envInt("SMGPU_PUSH_TIMEOUT_S", 60)) * 100000000ull;
    hipLaunchKernelGGL(k_flag_relay, dim3(1), dim3(64), 0, h->exch, writeIdx >= 0 ? h->dFlagWords + writeIdx : nullptr, tag,
                       waitIdx >= 0 ? (const unsigned*)(h->dFlagWords + waitIdx) : nullptr, tag, &h->st.acc->err, ticks);
    HIP_OK(hipGetLastError());
    return 0;
}
static bool mergedOk(const smgpu_handle* h) {
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    return h->mergedWanted && fused && h->useTiles && h->geomT == 256 && h->smoothT == 256 && h->nShared > 0 && h->shr.nTiles > 0 && h->nGeomShared > 0 &&
           h->packTiles && h->dMultiIdx && h->dPeer && h->dPosSlot && h->dRoleTickets && !h->layersOn && !h->bndOn && (!h->useExch || flagUsable(h)) && !h->geomAheadDone &&
           !h->st.inlinePackF &&
           // roleDone() counts one arrival per XCD slot: every role's workgroup count must be a multiple of 8 (tileGrid rounds up with the XCD map only)
           h->xcdMap != 0 &&
           // several engines on one device with the peer-store transport: workgroups that spin for a peer's flag hold their slots
           // while the peer's launches need some -- fine for a handful of tiles, not for a chip full of them
           !(h->pushOn && h->deviceShare > 1 && h->shr.nTiles > 256);
}
static bool flagged(const smgpu_handle* h) { return h->mergedIter && h->useExch; }
static int runMergedGeomPack(smgpu_handle* h) {
    const MeshView& m = h->mv;
    State s = h->st;
    if (flagged(h)) s.push = h->flagView;
    // the pack role goes behind the geometry tiles with a shared point and a first batch of the others: by the time its workgroups
    // are dispatched (the second generation of the launch) the first role has finished, so they hardly spin
    const int slots = 4 * 256;      // (4 workgroups per CU: k_geom_halo's launch bounds)
    const int gS = tileGrid(h->nGeomShared, h->xcdMap);
    // (flagged arrangement: the exchange stream's chain -- pack, exchange A, flag, shared points' role, exchange F, flag -- is what
    // bounds the iteration, so the pack role goes right behind the first role and spins for it)
    const int nI1 = std::min(h->nGeomInterior, std::max(0, envInt("SMGPU_HALO_PACK_AFTER", flagged(h) ? 0 : slots + slots / 2) - gS) & ~7);
    const int g1 = tileGrid(nI1, h->xcdMap), g2 = tileGrid(h->nGeomInterior - nI1, h->xcdMap), gP = tileGrid(h->shr.nTiles, h->xcdMap);
    HaloG hg;
    hg.geomS = h->dGeomShared; hg.nGeomS = h->nGeomShared; hg.geomI = h->dGeomInterior; hg.nGeomI = h->nGeomInterior; hg.nI1 = nI1;
    hg.nPack = h->shr.nTiles;
    hg.ticket = h->dRoleTickets; hg.serial = ++h->roleLaunches[0];
    hg.tagA = (unsigned)(h->haloIter + 1);
    hg.debug = envInt("SMGPU_HALO_DEBUG", 0);
    const PackView pk{h->dOwnA, h->sendA, h->dSendOff, h->dSendSlots, 0};
    const size_t lds = std::max(h->geomLds, h->haloLds);
    const int rc = launchKDispatch(h, K_GEOM_TILE, [&](hipEvent_t evA, hipEvent_t evB) {
        if (h->foamOrg)
            hipExtLaunchKernelGGL((k_geom_halo<256, true>), dim3(gS + g1 + gP + g2), dim3(256), (uint32_t)lds, h->stream, evA, evB, 0, m, s, h->gv, h->writeFaces ? 1 : 0, hg,
                                  h->xcdMap, h->deferN, h->deferIter, h->deferLocal, h->deferHist, h->hv, pk);
        else
            hipExtLaunchKernelGGL((k_geom_halo<256, false>), dim3(gS + g1 + gP + g2), dim3(256), (uint32_t)lds, h->stream, evA, evB, 0, m, s, h->gv, h->writeFaces ? 1 : 0, hg,
                                  h->xcdMap, h->deferN, h->deferIter, h->deferLocal, h->deferHist, h->hv, pk);
    });
    h->deferN = 0;
    h->deferLocal = h->deferHist = nullptr;
    return rc;
}
static int runMergedSmooth(smgpu_handle* h) {
    const Prm prm = makePrm(h);
    HaloS hs;
    hs.nTiles = h->stl.nTiles; hs.nSp = h->shr.nTiles;
    hs.nMulti = h->nMulti; hs.nMultiBlocks = h->nMulti ? gridFor((int64_t)h->nMulti * 16) : 0;
    hs.gM = ((hs.nMultiBlocks + 7) / 8) * 8;
    hs.multiIdx = h->dMultiIdx; hs.multiSlots = h->dMultiSlots; hs.combA = h->dCombA;
    hs.ticket = h->dRoleTickets + kRoleWords;
    hs.serial = hs.nMultiBlocks ? ++h->roleLaunches[1] : 0u;
    hs.pwA = pushWaitOf(h, 0); hs.tagF = (unsigned)(h->haloIter + 1);
    // k_shared_fix's work as a role of this launch, where exchange F completes while the launch runs: the peers' kernels move the
    // flags (peer stores; one rank of eight: 120.6 -> 114.6 us).  In the middle of the regular tiles: by then the shared points'
    // role (first in the launch) is through and exchange F has had ~20 us.  NOT with the flagged arrangement by default: there
    // exchange F is a chain on the exchange stream (flag, exchange kernel, flag: ~35 us behind the shared points' role) that ends
    // about when the launch does -- a role inside would stretch the launch (measured 124-127 against 122-123 us with k_shared_fix
    // behind it; SMGPU_HALO_FIX_INSIDE=1 forces the role).
    HaloFix hf{};
    h->fixInSmooth = envInt("SMGPU_HALO_FIX_INSIDE", h->pushOn ? 1 : 0) != 0 && (h->pushOn || flagged(h));
    hs.nA = h->stl.nTiles;
    if (h->fixInSmooth) {
        hs.nA = (int)((int64_t)h->stl.nTiles * std::min(100, std::max(0, envInt("SMGPU_HALO_FIX_AT", flagged(h) ? 85 : 55))) / 100) & ~7;
        hf.nFix = ((gridFor(h->nShared) + 7) / 8) * 8;
        hf.nShared = h->nShared; hf.sharedLocal = h->dSharedLocal; hf.combOff = h->dCombOff; hf.combSlots = h->dCombSlots; hf.recvF = h->recvF;
        hf.partialBase = h->stl.nTiles;
        hf.pwF = pushWaitOf(h, 1);
        hf.ticket = h->dRoleTickets + 2 * kRoleWords; hf.serial = ++h->roleLaunches[2];
        hf.sysLoads = flagged(h) ? 1 : 0;
    }
    const int grid = hs.gM + tileGrid(h->shr.nTiles, h->xcdMap) + tileGrid(hs.nA, h->xcdMap) + hf.nFix + tileGrid(h->stl.nTiles - hs.nA, h->xcdMap);
    const size_t lds = std::max(h->smoothLds, h->haloLds);
    State s = h->st;
    if (flagged(h)) s.push = h->flagView;
    s.ownF = h->fixInSmooth ? h->dOwnF : nullptr;
    return launchKDispatch(h, K_SMOOTH_FINAL, [&](hipEvent_t evA, hipEvent_t evB) {
        hipExtLaunchKernelGGL((k_smooth_halo<256>), dim3(grid), dim3(256), (uint32_t)lds, h->stream, evA, evB, 0, h->mv, s, prm, h->sv, h->hv, hs, h->xcdMap, hf);
    });
}

int smgpu_iter_begin(smgpu_handle* h) {
    if (!h || !h->haloOn) return fail("halo not configured");
    if (!h->prmSet) return fail("smgpu_set_params has not been called");
    HIP_OK(hipSetDevice(h->device));
    if (h->pushOn && h->pushStride != (h->st.lStride > 0 ? h->st.lStride : SMGPU_HALO_L_LAYERS)) {   // the L records grew (boundary set-up)
        HIP_OK(hipStreamSynchronize(h->stream));
        if (pushBuildTables(h)) return 1;
    }
    h->st.stepSqr = swapApplyHalo(h) ? h->dStepSqr : nullptr;      // (decided per iteration: parameters and the optional features may change between calls)
    h->mergedIter = mergedOk(h);
    if (h->mergedIter) {      // geometry and exchange A's pack in one launch
        if (h->useExch && ensureFlagView(h)) return 1;
        if (runMergedGeomPack(h)) return 1;
        if (updateWalkMode(h)) return 1;
        if (h->useExch) return flagRelay(h, -1, 0);      // the host's exchange A may start as soon as the pack ROLE is through (not the launch)
        return exchAfterCompute(h);
    }
    if (runBndPre(h)) return 1;
    if (h->geomAheadDone) {
        // the tiles away from the shared points were recomputed by smgpu_iter_ahead of the previous iteration
        if (runGeometry(h, h->dGeomShared, h->nGeomShared, false, true)) return 1;
        h->geomAheadDone = false;
    } else if (runGeometry(h, nullptr, 0, false, true)) return 1;
    // a rank without shared geometry tiles launched nothing above: the previous iteration's reduction, parked for that
    // launch by smgpu_iter_end, must not wait for the next one (smgpu_iter_mid overwrites the partials before it)
    if (flushDeferred(h)) return 1;
    if (updateWalkMode(h)) return 1;
    if (forkFaFilter(h)) return 1;
    State s = h->st;
    const MeshView& m = h->mv;
    // boundary point smoothing: local normal sums and feature edge projections of the current coordinates (SM.C:2266, BPS.C:866),
    // started next to the geometry kernel above when there is a side stream
    if (h->bndOn) {
        if (h->bndPreDone) h->bndPreDone = false;
        else if (h->bndPreInFlight) {
            if (depWait(h, DEP_BND_JOIN, h->stream, h->evBndJoin)) return 1;
            h->bndPreInFlight = false;
        } else if (launchBndPre(h, m, s, h->stream)) return 1;
    }
    if (h->nShared)
        if (launchK(h, K_HALO, [&] {
                const bool withL = h->layersOn || h->bndOn;   // local normals / neighbour coordinates / feature projections (SM.C:2266, 2286, 2310-2330)
                const unsigned tag = (unsigned)(h->haloIter + 1);   // the peer-store transport's flag value for this iteration
                const PackLArgs la{h->dSharedLocal, h->dOwnL, h->nShared, h->dSendOff, h->dSendSlots, h->sendL, h->bv.bfOff, h->bv.inner, h->bv.featOfBnd,
                                   h->bv.featSum, h->bv.featCnt};
                if (h->useTiles && h->nSharedTiles > 0 && h->packTiles) {   // the staged gather of the smoothing tiles
                    const PackView pk{h->dOwnA, h->sendA, h->dSendOff, h->dSendSlots, h->bndOn ? 1 : 0};
                    // exchange L's records are packed by the first workgroups of the same launch
                    const int T = h->smoothT, nL = withL ? (((h->nShared + T - 1) / T + 7) / 8) * 8 : 0;
                    const dim3 grid(nL + tileGrid(h->nSharedTiles, h->xcdMap));
                    if (T == 64) hipLaunchKernelGGL(k_pack_tile<64>, grid, dim3(64), h->smoothLds, h->stream, m, s, h->sv, pk, h->dSharedTiles, h->nSharedTiles, h->xcdMap, la, nL, tag);
                    else if (T == 128) hipLaunchKernelGGL(k_pack_tile<128>, grid, dim3(128), h->smoothLds, h->stream, m, s, h->sv, pk, h->dSharedTiles, h->nSharedTiles, h->xcdMap, la, nL, tag);
                    else hipLaunchKernelGGL(k_pack_tile<256>, grid, dim3(256), h->smoothLds, h->stream, m, s, h->sv, pk, h->dSharedTiles, h->nSharedTiles, h->xcdMap, la, nL, tag);
                } else {
                    hipLaunchKernelGGL(k_halo_packA, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, m, s, h->dSharedLocal, h->dOwnA, h->nShared,
                                       h->dSendOff, h->dSendSlots, h->sendA, h->bndOn ? 1 : 0);
                    if (withL) hipLaunchKernelGGL(k_halo_packL, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, s, la);
                }
            })) return 1;
    return exchAfterCompute(h);             // sendA (and sendL) complete: the exchange may start
}

// Work of the iteration that does not depend on exchange A: the proposal (and, with the constraints off,
// the final move) of every tile that holds no shared point.  Optional: smgpu_iter_mid does it when skipped.
// With an exchange stream configured its kernels run next to exchange A.
int smgpu_iter_interior(smgpu_handle* h) {
    if (!h || !h->haloOn) return fail("halo not configured");
    HIP_OK(hipSetDevice(h->device));
    // without an exchange stream nothing runs next to the exchange: splitting the launch would only add ramp-up and
    // drain time (measured on 100^3 with 30 k shared points: 2 x 32.7 us instead of 45 us), so smgpu_iter_mid does it all
    if (!h->useTiles || h->interiorDone || !h->useExch || h->mergedIter) return 0;
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    const Prm prm = makePrm(h);
    if (fused) { if (runSmooth<true>(h, h->mv, h->st, prm, h->dInteriorTiles, h->nInteriorTiles)) return 1; }
    else if (runSmooth<false>(h, h->mv, h->st, prm, h->dInteriorTiles, h->nInteriorTiles)) return 1;
    h->interiorDone = true;
    return 0;
}

int smgpu_iter_mid(smgpu_handle* h) {
    if (!h || !h->haloOn) return fail("halo not configured");
    HIP_OK(hipSetDevice(h->device));
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    const Prm prm = makePrm(h);
    if (h->mergedIter) {      // combine, smoothing and exchange F's pack in one launch
        if (h->useExch) {
            // exchange A has been enqueued on the exchange stream: the word the shared points' role polls goes up behind it, and the
            // same wave waits for that role's "exchange F is packed" (exchange F is enqueued behind it)
            // (the kernel first: were its launch refused, a relay already enqueued would spin on the high-priority exchange stream
            // until its time-out with everything that synchronises that stream behind it; the exchange stream's order is the same)
            if (runMergedSmooth(h)) return 1;
            return flagRelay(h, 16, 1);
        }
        if (runMergedSmooth(h)) return 1;
        return exchAfterCompute(h);
    }
    if (h->useTiles && !h->interiorDone && smgpu_iter_interior(h)) return 1;
    if (computeAfterExch(h)) return 1;      // exchange A has been enqueued by the host
    if (h->nShared)
        if (launchK(h, K_HALO, [&] {
                // (with inlineCombine only the workgroups of the points with more than two sharers: the others are the smoothing kernel's)
                const int nTwo = h->st.inlineCombine ? 0 : gridFor(h->nShared), nMultiBlocks = h->nMulti ? gridFor((int64_t)h->nMulti * 16) : 0;
                const bool withL = h->layersOn || h->bndOn;
                if (withL && h->dMultiIdx && h->dPeer) {
                    // exchange A's and exchange L's combines and the shared boundary normals (OBB.C:201-230) in one launch
                    hipLaunchKernelGGL(k_halo_combineAL, dim3(nTwo + nMultiBlocks + gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->nShared, h->dPeer, h->dOwnA,
                                       h->recvA, h->dCombA, nTwo, h->nMulti, h->dMultiIdx, h->dMultiSlots, nTwo + nMultiBlocks, h->st, h->bv, h->bndOn ? 1 : 0,
                                       h->dCombOff, h->dCombSlots, h->dOwnL, h->recvL, h->dCombL, h->dSharedLocal, pushWaitOf(h, 0));
                    return;
                }
                if (nTwo + nMultiBlocks > 0 && h->dMultiIdx && h->dPeer)
                    hipLaunchKernelGGL(k_halo_combineA2, dim3(nTwo + nMultiBlocks), dim3(kBlock), 0, h->stream, h->nShared, h->dPeer, h->dOwnA, h->recvA, h->dCombA,
                                       nTwo, h->nMulti, h->dMultiIdx, h->dMultiSlots, pushWaitOf(h, 0), h->st.ownFold);
                else if (nTwo + nMultiBlocks > 0)
                    hipLaunchKernelGGL(k_halo_combineA, dim3(nTwo + nMultiBlocks), dim3(kBlock), 0, h->stream, h->nShared, h->dCombOff, h->dCombSlots,
                                       h->dOwnA, h->recvA, h->dCombA, &h->st.acc->err, h->dMultiIdx ? 1 : 0, nTwo, h->nMulti, h->dMultiIdx, h->dMultiSlots,
                                       pushWaitOf(h, 0), h->st.ownFold);
                if (withL)
                    hipLaunchKernelGGL(k_halo_combineL, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->nShared, h->dCombOff,
                                       h->dCombSlots, h->dOwnL, h->recvL, h->dCombL, h->st.lStride, h->st.ownFold);
                if (h->bndOn)   // OBB.C:201-230 for the shared boundary points, on the sums
                    hipLaunchKernelGGL(k_bnd_normals_shared, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->st, h->bv, h->nShared, h->dSharedLocal);
            })) return 1;
    if (h->useTiles) {
        // the tiles with shared points only when the others have been done next to the exchange, otherwise all of them
        const int* list = h->interiorDone ? h->dSharedTiles : nullptr;
        const int nList = h->interiorDone ? h->nSharedTiles : 0;
        if (h->interiorDone && nList == 0) { /* every tile has been done */ }
        else if (fused) { if (runSmooth<true>(h, h->mv, h->st, prm, list, nList)) return 1; }
        else if (runSmooth<false>(h, h->mv, h->st, prm, list, nList)) return 1;
        // the boundary points the smoothing kernels skipped; their partials follow the tiles' and the shared points' (k_shared_fix)
        if (h->bndOn) { if (fused ? launchBndFix<true>(h, h->stl.nTiles + gridFor(h->nShared)) : launchBndFix<false>(h, 0)) return 1; }
        if (!fused && runConstraints(h)) return 1;
    } else if (runProposalAndConstraints(h)) return 1;
    if (h->nSend && !(fused && h->st.inlinePackF && !h->bndOn && h->useTiles))   // (fused: the smoothing kernel wrote the flags into the send slots)
        if (launchK(h, K_HALO, [&] {
                hipLaunchKernelGGL(k_halo_packF, dim3(gridFor(h->nSend)), dim3(kBlock), 0, h->stream, h->nSend, h->dSendShared,
                                   h->dSharedLocal, h->st.frozen, h->sendF, h->st.push, (unsigned)(h->haloIter + 1));
            })) return 1;
    return exchAfterCompute(h);             // sendF is complete: exchange F may start
}

// Optional, between smgpu_iter_mid and smgpu_iter_end (constraints off): while exchange F is in flight, start
// the NEXT iteration's geometry for every tile whose cells hold no shared point -- their points are final.
int smgpu_iter_ahead(smgpu_handle* h) {
    if (!h || !h->haloOn) return fail("halo not configured");
    HIP_OK(hipSetDevice(h->device));
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    if (!h->useTiles || !fused || h->geomAheadDone || !h->useExch || h->mergedIter) return 0;   // see smgpu_iter_interior
    if (runGeometry(h, h->dGeomInterior, h->nGeomInterior, true)) return 1;
    h->geomAheadDone = true;
    return 0;
}

int smgpu_iter_end(smgpu_handle* h) {
    if (!h || !h->haloOn) return fail("halo not configured");
    HIP_OK(hipSetDevice(h->device));
    const MeshView& m = h->mv;
    State s = h->st;
    const Prm prm = makePrm(h);
    const bool fused = !h->prm.edgeAngleConstraint && !h->prm.faceAngleConstraint;
    const bool fusedTiles = fused && h->useTiles;
    const bool flg = flagged(h);
    const bool swapHalo = !fusedTiles && s.stepSqr != nullptr;      // (as smgpu_iter_begin decided for this iteration)
    if (flg) { if (flagRelay(h, 17, -1)) return 1; }      // behind exchange F: the fix role / k_shared_fix polls it
    else if (computeAfterExch(h)) return 1;      // exchange F has been enqueued by the host
    s.stats = nullptr;                      // per-iteration results go to localStats in this mode
    int nPart;
    if (fusedTiles) {
        // every non-shared point is already final; finish the shared ones (or of the freeze flags included)
        const int gS = gridFor(h->nShared);
        if (h->nShared && !(h->mergedIter && h->fixInSmooth))      // (else: the fix role of k_smooth_halo has done it)
            if (launchK(h, K_HALO, [&] {
                    hipLaunchKernelGGL(k_shared_fix, dim3(gS), dim3(kBlock), 0, h->stream, m, s, prm, h->nShared, h->dSharedLocal, h->dCombOff,
                                       h->dCombSlots, h->recvF, h->stl.nTiles, pushWaitOf(h, 1), flg ? 1 : 0);
                })) return 1;
        nPart = h->stl.nTiles + (h->bndOn ? gridFor(h->nShared) + gridFor(2 * (int64_t)h->bv.nB) : (h->nShared ? gS : 0));
    } else {
        if (h->nShared && h->nRecv)
            if (launchK(h, K_HALO, [&] {
                    hipLaunchKernelGGL(k_halo_orF, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->nShared, h->dSharedLocal,
                                       h->dCombOff, h->dCombSlots, h->recvF, h->st.frozen, pushWaitOf(h, 1));
                })) return 1;
        if (swapHalo) {
            const int gSwap = (int)((m.nPoints + (int64_t)kApplyPer * kBlock - 1) / ((int64_t)kApplyPer * kBlock));
            if (launchK(h, K_APPLY, [&] { hipLaunchKernelGGL(k_apply_swap, dim3(gSwap), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
            nPart = gSwap;
        } else {
            if (launchK(h, K_APPLY, [&] { hipLaunchKernelGGL(k_apply, dim3(gridFor(m.nPoints)), dim3(kBlock), 0, h->stream, m, s, prm); })) return 1;
            nPart = gridFor(m.nPoints);
        }
    }
    // relTol = -1: the stop decision needs the all-rank residual and is the host's (SM.C:1567,2401)
    double* hist = (h->statsHistory && h->statsHistoryCap > 0) ? h->statsHistory + 2 * (size_t)(h->statsHistoryN++ % h->statsHistoryCap) : nullptr;
    // with a stats history nobody reads the record before the loop ends: the reduction then rides in the next geometry
    // launch (first launch of smgpu_iter_begin) as in smgpu_iterate; flushDeferred closes the last iteration
    // (constraints on: the partials are k_apply_swap's, as in smgpu_iterate -- nothing between here and the next geometry launch
    // reads the counters the reduction resets)
    if (hist && (fusedTiles || swapHalo) && h->geomT >= 64 && envInt("SMGPU_DEFER_FINISH", 1)) {
        h->deferN = nPart; h->deferIter = -1 /* no stats[] record in this mode */; h->deferLocal = h->localStats; h->deferHist = hist;
    } else if (launchK(h, K_FINISH, [&] { hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishBlock), 0, h->stream, s, nPart, h->haloIter, -1.0, h->localStats, hist); })) return 1;
    if (swapHalo) std::swap(h->st.ptsCur, h->st.prop);      // the proposal array is the next coordinates (k_apply_swap restored the points that stay)
    else std::swap(h->st.ptsCur, h->st.ptsNext);
    h->haloIter++;
    h->interiorDone = false;
    // (flagged arrangement with a stats history: nobody reads localStats per iteration, and an exchange stream that waited for the
    // whole iteration here could not start the next exchange A next to the next geometry launch)
    if (flg && hist) return 0;
    return exchAfterCompute(h);             // localStats is complete: the host may reduce / copy it
}

// ---- optional boundary layer treatment -------------------------------------------------------------------------
int smgpu_layers_begin(smgpu_handle* h, const smgpu_layer_desc* d, int32_t* enabled, int32_t* maxIter) {
    if (!h || !d) return fail("null argument");
    if (d->nPatches < 0 || (d->nPatches && (!d->patchStart || !d->patchSize || !d->patchKind || !d->isLayerPatch))) return fail("bad patch description");
    HIP_OK(hipSetDevice(h->device));
    bool any = false;
    std::vector<LayerPatch> patches((size_t)d->nPatches);
    for (int i = 0; i < d->nPatches; ++i) {
        patches[(size_t)i] = LayerPatch{d->patchStart[i], d->patchSize[i], (int32_t)d->patchKind[i], d->isLayerPatch[i] != 0};
        any = any || d->isLayerPatch[i];
    }
    const bool on = any && (d->layerMaxBlendingFraction > 1.0e-15);   // SM.C:2025, SMALL
    if (enabled) *enabled = on ? 1 : 0;
    if (maxIter) *maxIter = d->maxLayers + 1;
    h->layersOn = false;
    h->lb.reset();
    if (!on) return 0;
    // face area vectors of the current coordinates (fvPatch::Sf): the direct face kernel writes all of them
    const MeshView& m = h->mv;
    State s = h->st;
    hipLaunchKernelGGL(k_face_geom, dim3(gridFor(m.nFaces)), dim3(kBlock), 0, h->stream, m, s, 0, h->foamOrg ? 1 : 0);
    h->lbArea.resize(3 * (size_t)m.nFaces);
    HIP_OK(hipMemcpyAsync(h->lbArea.data(), s.fArea, sizeof(double) * h->lbArea.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    h->lbInternal.resize((size_t)m.nPoints);
    HIP_OK(hipMemcpy(h->lbInternal.data(), m.pflags, (size_t)m.nPoints, hipMemcpyDeviceToHost));
    for (auto& f : h->lbInternal) f = (f & PF_INTERNAL) ? 1 : 0;
    h->lb.reset(new LayerBuilder());
    const std::string err = h->lb->begin(h->topo, h->lbInternal.data(), patches, h->lbArea.data(), d->layerMaxBlendingFraction, d->layerEdgeLength,
                                         d->layerExpansionRatio, d->minLayers, d->maxLayers);
    if (!err.empty()) { h->lb.reset(); return fail(err); }
    return 0;
}

int smgpu_layers_step(smgpu_handle* h, int32_t step, int32_t arg) {
    if (!h) return fail("null handle");
    if (!h->lb) return fail("smgpu_layers_step: no set-up in progress (smgpu_layers_begin, enabled)");
    LayerBuilder& b = *h->lb;
    switch (step) {
    case SMGPU_LAYERS_HOPS_SWEEP: b.hopsSweep(); return 0;
    case SMGPU_LAYERS_NORMALS_ACCUMULATE: b.normalsAccumulate(); return 0;
    case SMGPU_LAYERS_NORMALS_FINISH: b.normalsFinish(); return 0;
    case SMGPU_LAYERS_PROPAGATE_SWEEP: b.propagateSweep(arg); return 0;
    case SMGPU_LAYERS_FINISH: break;
    default: return fail("smgpu_layers_step: unknown step");
    }
    HIP_OK(hipSetDevice(h->device));
    b.finish();
    const LayerSetup& ls = b.out;
    const double* dn = nullptr; const int *dh = nullptr, *dm = nullptr; const double *dl = nullptr, *db = nullptr;
    if (devUpload(h, &dn, ls.normals) || devUpload(h, &dh, ls.hops) || devUpload(h, &dm, ls.outerMap) || devUpload(h, &dl, ls.lengthOfHops) ||
        devUpload(h, &db, ls.blendOfHops)) return 1;
    h->st.layerNormal = const_cast<double*>(dn);
    h->st.layerHops = dh; h->st.layerMap = dm; h->st.layerLen = dl; h->st.layerBlend = db;
    h->layerHopsHost = ls.hops;
    h->layerMapHost = ls.outerMap;
    h->st.combL = nullptr;
    if (h->haloOn) {
        if (h->nSend && (!h->sendL || !h->recvL)) return fail("boundary layer treatment with a halo needs smgpu_halo_desc.sendL / recvL");
        if (devAlloc(h, &h->dOwnL, (size_t)std::max(h->nShared, 1) * SMGPU_HALO_L_DOUBLES)) return 1;
        if (devAlloc(h, &h->dCombL, (size_t)std::max(h->nShared, 1) * SMGPU_HALO_L_DOUBLES)) return 1;
        h->st.combL = h->dCombL;
    }
    h->lb.reset();
    h->lbArea.clear(); h->lbArea.shrink_to_fit();
    h->layersOn = true;
    return 0;
}

int smgpu_layers_shared(smgpu_handle* h, int32_t field, int32_t set, double* v) {
    if (!h || !v) return fail("null argument");
    if (!h->lb) return fail("smgpu_layers_shared: no set-up in progress");
    if (!h->haloOn) return fail("smgpu_layers_shared: halo not configured");
    LayerSetup& o = h->lb->out;
    const std::vector<int>& sl = h->sharedLocalHost;
    const size_t n = sl.size();
    for (size_t i = 0; i < n; ++i) {
        const size_t p = (size_t)sl[i];
        if (field == SMGPU_LAYERS_F_HOPS) {
            if (set) o.hops[p] = (int32_t)v[i]; else v[i] = o.hops[p];
        } else if (field == SMGPU_LAYERS_F_NORMALS_COUNT || field == SMGPU_LAYERS_F_NORMALS) {
            const size_t w = field == SMGPU_LAYERS_F_NORMALS_COUNT ? 4 : 3;
            for (size_t c = 0; c < 3; ++c) { if (set) o.normals[3 * p + c] = v[w * i + c]; else v[w * i + c] = o.normals[3 * p + c]; }
            if (w == 4) { if (set) h->lb->nFaces[p] = (int32_t)v[4 * i + 3]; else v[4 * i + 3] = h->lb->nFaces[p]; }
        } else return fail("smgpu_layers_shared: unknown field");
    }
    return 0;
}

int smgpu_set_layers(smgpu_handle* h, const smgpu_layer_desc* d, int32_t* enabled) {
    if (!h || !d) return fail("null argument");
    if (h->haloOn) return fail("smgpu_set_layers is the serial set-up; with a halo use smgpu_layers_begin / step / shared");
    int32_t on = 0, maxIter = 0;
    if (smgpu_layers_begin(h, d, &on, &maxIter)) return 1;
    if (enabled) *enabled = on;
    if (!on) return 0;
    for (int i = 0; i < maxIter; ++i) h->lb->hopsSweep();
    h->lb->normalsAccumulate();
    h->lb->normalsFinish();
    for (int i = 1; i <= maxIter; ++i) h->lb->propagateSweep(i);
    return smgpu_layers_step(h, SMGPU_LAYERS_FINISH, 0);
}

// ---- optional boundary point smoothing ---------------------------------------------------------------------------
// ---- the set-up in steps (serial: smgpu_set_boundary_smoothing runs them back to back) ------------------------------
// host copies the set-up works on
static int bndFetchHost(smgpu_handle* h) {
    const int P = h->topo.nPoints;
    h->bndPts.resize(3 * (size_t)P);
    HIP_OK(hipMemcpyAsync(h->bndPts.data(), h->st.ptsCur, sizeof(double) * h->bndPts.size(), hipMemcpyDeviceToHost, h->stream));
    h->bndFlags.resize((size_t)P);
    h->bndInternal.resize((size_t)P);
    HIP_OK(hipMemcpyAsync(h->bndFlags.data(), h->mv.pflags, (size_t)P, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    for (int p = 0; p < P; ++p) h->bndInternal[(size_t)p] = (h->bndFlags[(size_t)p] & PF_INTERNAL) ? 1 : 0;
    return 0;
}

int smgpu_boundary_stats(smgpu_handle* h, double* minEdgeLength, double* bb) {
    if (!h || !minEdgeLength || !bb) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    if (bndFetchHost(h)) return 1;
    const Topology& t = h->topo;
    const std::vector<double>& pts = h->bndPts;
    // getMeshStats SM.C:1478-1526: minimum edge length and the bounding box of the edge end points (min x, max x, min y, ...)
    double shortest = 1.0e300;
    for (int c = 0; c < 3; ++c) { bb[2 * c] = 1.0e300; bb[2 * c + 1] = -1.0e300; }
    for (int e = 0; e < t.nEdges; ++e) {
        const double* a = &pts[3 * (size_t)t.edges[2 * (size_t)e]];
        const double* b = &pts[3 * (size_t)t.edges[2 * (size_t)e + 1]];
        const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
        const double len = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (len < shortest) shortest = len;
        for (int c = 0; c < 3; ++c) {
            if (a[c] < bb[2 * c]) bb[2 * c] = a[c];
            if (a[c] > bb[2 * c + 1]) bb[2 * c + 1] = a[c];
            if (b[c] < bb[2 * c]) bb[2 * c] = b[c];
            if (b[c] > bb[2 * c + 1]) bb[2 * c + 1] = b[c];
        }
    }
    *minEdgeLength = shortest;
    return 0;
}

int smgpu_boundary_begin(smgpu_handle* h, const smgpu_boundary_desc* d, double minEdgeLengthGlobal, double perimeterGlobal,
                         smgpu_boundary_info* info) {
    if (!h || !d) return fail("null argument");
    if (d->nPatches < 0 || (d->nPatches && (!d->patchStart || !d->patchSize || !d->patchKind || !d->isSmoothingPatch))) return fail("bad patch description");
    if ((d->nInitEdges && (!d->initEdges || !d->initEdgePoints)) || (d->nTargetEdges && (!d->targetEdges || !d->targetEdgePoints)) ||
        (d->nSurfaceTriangles && (!d->surfaceTriangles || !d->surfacePoints)))
        return fail("smgpu_boundary_begin: null geometry array");
    HIP_OK(hipSetDevice(h->device));
    const Topology& t = h->topo;
    h->bndOn = false;
    h->bndPending = false;
    if (info) std::memset(info, 0, sizeof(*info));
    if (bndFetchHost(h)) return 1;

    BoundaryInputHost in;
    auto fillEdges = [](EdgeMeshHost& em, int nP, const double* ep, int nE, const int32_t* e) {
        if (nP > 0 && ep) em.pts.assign(ep, ep + 3 * (size_t)nP);
        if (nE > 0 && e) em.edges.assign(e, e + 2 * (size_t)nE);
    };
    fillEdges(in.initEdges, d->nInitEdgePoints, d->initEdgePoints, d->nInitEdges, d->initEdges);
    fillEdges(in.targetEdges, d->nTargetEdgePoints, d->targetEdgePoints, d->nTargetEdges, d->targetEdges);
    h->bndSurfPts.clear(); h->bndSurfTris.clear();
    if (d->nSurfaceTriangles > 0) {
        h->bndSurfPts.assign(d->surfacePoints, d->surfacePoints + 3 * (size_t)d->nSurfacePoints);
        h->bndSurfTris.assign(d->surfaceTriangles, d->surfaceTriangles + 3 * (size_t)d->nSurfaceTriangles);
    }
    in.surfPts = h->bndSurfPts;
    in.surfTris = h->bndSurfTris;
    in.isCornerPointIO = d->isCornerPointIO;
    in.isFeatureEdgePointIO = d->isFeatureEdgePointIO;
    in.distanceTolerance = d->distanceTolerance;
    in.meshMinEdgeLength = minEdgeLengthGlobal;
    in.meshPerimeter = perimeterGlobal;
    h->bndPatches.assign((size_t)d->nPatches, BndPatch{});
    for (int i = 0; i < d->nPatches; ++i) h->bndPatches[(size_t)i] = BndPatch{d->patchStart[i], d->patchSize[i], (int32_t)d->patchKind[i], d->isSmoothingPatch[i] != 0};
    h->bndBlend = d->internalSmoothingBlendingFraction;
    BoundarySetup& bs = h->bs;
    const std::string err = buildBoundarySetup(t, h->bndInternal.data(), h->bndPts.data(), h->bndPatches, in, bs);
    if (!err.empty()) return fail(err);
    if (info) {
        info->enabled = bs.enabled ? 1 : 0;
        info->nCornerPoints = bs.nCorner; info->nFeatureEdgePoints = bs.nFeature;
        info->nSmoothingSurfacePoints = bs.nSmoothingSurface; info->nFrozenSurfacePoints = bs.nFrozenSurface;
        int mx = -1;
        for (int32_t v : bs.targetEdgeStrings) mx = std::max(mx, (int)v);
        info->nTargetEdgeStrings = mx + 1;
    }
    h->bndPending = bs.enabled;
    return 0;
}

// device tables over the boundary (non-internal) points; needs the final hop counts
static int bndTables(smgpu_handle* h) {
    const Topology& t = h->topo;
    const MeshView& m = h->mv;
    const int P = t.nPoints;
    BoundarySetup& bs = h->bs;
    const std::string err = boundarySetupFinish(t, h->bndPts.data(), bs);
    if (!err.empty()) return fail(err);
    const std::vector<uint8_t>& internal = h->bndInternal;
    std::vector<int> bpts, inner, featPts, featString, featOfBnd, bfOff(1, 0), bfVal;
    std::vector<uint8_t> flags, ptClass((size_t)P, 0);
    std::vector<double> corner;
    std::vector<int> bndOf((size_t)P, -1);
    for (int p = 0; p < P; ++p) {
        if (internal[(size_t)p]) continue;
        bndOf[(size_t)p] = (int)bpts.size();
        bpts.push_back(p);
        uint8_t f = 0;
        if (bs.isCornerPoint[(size_t)p]) f |= BF_CORNER;
        if (bs.isFeatureEdgePoint[(size_t)p]) f |= BF_FEATURE;
        if (bs.isSmoothingSurfacePoint[(size_t)p]) f |= BF_SMOOTHSURF;
        if (bs.isConnectedToInternalPoint[(size_t)p]) f |= BF_CONNECTED;
        flags.push_back(f);
        ptClass[(size_t)p] = f & (BF_CORNER | BF_FEATURE);
        inner.push_back(bs.innerMap[(size_t)p]);
        for (int c = 0; c < 3; ++c) corner.push_back(bs.cornerPoints[3 * (size_t)p + c]);
        if (f & BF_FEATURE) { featOfBnd.push_back((int)featPts.size()); featPts.push_back(p); featString.push_back(bs.pointStrings[(size_t)p]); }
        else featOfBnd.push_back(-1);
    }
    const int nB = (int)bpts.size();
    {   // boundary faces per boundary point on ordinary patches (OBB.C:156-159), ascending face id = the reference's order
        std::vector<std::vector<int>> bf((size_t)nB);
        for (const BndPatch& pp : h->bndPatches) {
            if (pp.kind != 0) continue;
            for (int f = pp.start; f < pp.start + pp.size; ++f)
                for (int k = t.facePoints.off[f]; k < t.facePoints.off[f + 1]; ++k) {
                    const int bi = bndOf[(size_t)t.facePoints.val[k]];
                    if (bi >= 0) bf[(size_t)bi].push_back(f);
                }
        }
        for (int i = 0; i < nB; ++i) {
            std::sort(bf[(size_t)i].begin(), bf[(size_t)i].end());
            bfVal.insert(bfVal.end(), bf[(size_t)i].begin(), bf[(size_t)i].end());
            bfOff.push_back((int)bfVal.size());
        }
    }
    Bvh bvh;
    bvh.build(h->bndSurfPts, h->bndSurfTris);
    if (7 * bvh.wideDepth + 1 > kBvhStack) return fail("boundary set-up: the target surface's hierarchy is deeper than the traversal stack");
    BndView& v = h->bv;
    v = BndView{};
    v.nB = nB;
    const int *dPts = nullptr, *dInner = nullptr, *dBfOff = nullptr, *dBfVal = nullptr, *dFeatPts = nullptr, *dFeatStr = nullptr, *dFeatOf = nullptr;
    const int *dTeE = nullptr, *dTeS = nullptr, *dRef = nullptr;
    const float* dBox = nullptr;
    const uint8_t *dFlags = nullptr, *dClass = nullptr;
    const double *dCorner = nullptr, *dTeP = nullptr, *dTri = nullptr;
    if (devUpload(h, &dPts, bpts) || devUpload(h, &dInner, inner) || devUpload(h, &dBfOff, bfOff) || devUpload(h, &dBfVal, bfVal) ||
        devUpload(h, &dFeatPts, featPts) || devUpload(h, &dFeatStr, featString) || devUpload(h, &dFeatOf, featOfBnd) ||
        devUpload(h, &dFlags, flags) || devUpload(h, &dClass, ptClass) || devUpload(h, &dCorner, corner) ||
        devUpload(h, &dTeP, bs.target.pts) || devUpload(h, &dTeE, bs.target.edges) || devUpload(h, &dTeS, bs.targetEdgeStrings) ||
        devUpload(h, &dBox, bvh.wideBox) || devUpload(h, &dRef, bvh.wideRef) || devUpload(h, &dTri, bvh.triVerts))
        return 1;
    v.pts = dPts; v.flags = const_cast<uint8_t*>(dFlags); v.corner = dCorner; v.inner = dInner; v.bfOff = dBfOff; v.bfVal = dBfVal;
    v.ptClass = dClass;
    v.nFeat = (int)featPts.size(); v.featPts = dFeatPts; v.featString = dFeatStr; v.featOfBnd = dFeatOf;
    if (devAlloc(h, &v.featSum, 3 * (size_t)std::max(v.nFeat, 1)) || devAlloc(h, &v.featCnt, (size_t)std::max(v.nFeat, 1))) return 1;
    v.nTE = bs.target.nEdges(); v.tePts = dTeP; v.teEdges = dTeE; v.teString = dTeS;
    {   // string -> edges
        int nStr = 0;
        for (int32_t sid : bs.targetEdgeStrings) nStr = std::max(nStr, (int)sid + 1);
        std::vector<int> strOff((size_t)nStr + 1, 0), strEdges(bs.targetEdgeStrings.size());
        for (int32_t sid : bs.targetEdgeStrings) if (sid >= 0) ++strOff[(size_t)sid + 1];
        for (int i = 0; i < nStr; ++i) strOff[(size_t)i + 1] += strOff[(size_t)i];
        std::vector<int> fill(strOff.begin(), strOff.end() - 1);
        for (size_t e = 0; e < bs.targetEdgeStrings.size(); ++e)
            if (bs.targetEdgeStrings[e] >= 0) strEdges[(size_t)fill[(size_t)bs.targetEdgeStrings[e]]++] = (int)e;
        const int *dSo = nullptr, *dSe = nullptr;
        if (devUpload(h, &dSo, strOff) || devUpload(h, &dSe, strEdges)) return 1;
        v.strOff = dSo; v.strEdges = dSe;
    }
    v.nNodes = (int)(bvh.wideRef.size() / 16); v.wideBox = dBox; v.wideRef = dRef; v.triVerts = dTri;
    v.distanceTolerance = bs.distanceTolerance;
    v.internalBlend = h->bndBlend;
    // isSmoothingSurfacePoint is this classification's from now on (BPS.C:404-412)
    for (int p = 0; p < P; ++p) {
        h->bndFlags[(size_t)p] &= (uint8_t)~PF_SMOOTHSURF;
        if (bs.isSmoothingSurfacePoint[(size_t)p]) h->bndFlags[(size_t)p] |= PF_SMOOTHSURF;
    }
    HIP_OK(hipMemcpy(const_cast<uint8_t*>(m.pflags), h->bndFlags.data(), (size_t)P, hipMemcpyHostToDevice));
    if (h->haloOn) {   // shared points: their place in the boundary tables, and the exchange-L staging buffers
        std::vector<int> bos(h->sharedLocalHost.size() + 1, -1);
        for (size_t i = 0; i < h->sharedLocalHost.size(); ++i) bos[i] = bndOf[(size_t)h->sharedLocalHost[i]];
        const int* dBos = nullptr;
        if (devUpload(h, &dBos, bos)) return 1;
        h->st.bndOfShared = dBos;
        if (h->nSend && (!h->sendL || !h->recvL)) return fail("boundary point smoothing with a halo needs smgpu_halo_desc.sendL / recvL");
        if (!h->dOwnL && devAlloc(h, &h->dOwnL, (size_t)std::max(h->nShared, 1) * SMGPU_HALO_L_DOUBLES)) return 1;
        if (!h->dCombL && devAlloc(h, &h->dCombL, (size_t)std::max(h->nShared, 1) * SMGPU_HALO_L_DOUBLES)) return 1;
        h->st.combL = h->dCombL;
        h->st.lStride = SMGPU_HALO_L_DOUBLES;   // from now on the boundary point smoothing fields travel too
    }
    h->bndNormalsFromLayers = h->st.layerNormal != nullptr;   // SM.C:2219 is done by the layer set-up when that ran
    if (!h->st.layerNormal) {
        if (devAlloc(h, &h->st.layerNormal, 3 * (size_t)P)) return 1;
        HIP_OK(hipMemsetAsync(h->st.layerNormal, 0, sizeof(double) * 3 * (size_t)P, h->stream));
    }
    if (!h->bndSide && envInt("SMGPU_SIDE_STREAM", sideStreamDefault(h->mv.nPoints))) {
        if (depInit(h)) return 1;
        if (hipStreamCreateWithFlags(&h->bndSide, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->evBndFork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->evBndJoin, hipEventDisableTiming) != hipSuccess)
            return fail("side stream creation failed");
    }
    h->bndOn = true;
    h->bndPending = false;
    h->bndPts.clear(); h->bndPts.shrink_to_fit();
    h->bndSurfPts.clear(); h->bndSurfPts.shrink_to_fit();
    h->bndSurfTris.clear(); h->bndSurfTris.shrink_to_fit();
    return 0;
}

int smgpu_boundary_step(smgpu_handle* h, int32_t step) {
    if (!h) return fail("null handle");
    HIP_OK(hipSetDevice(h->device));
    const MeshView& m = h->mv;
    switch (step) {
    case SMGPU_BOUNDARY_HOPS_SWEEP:
        if (!h->bndPending) return fail("smgpu_boundary_step: no set-up in progress (smgpu_boundary_begin, enabled)");
        boundarySetupHopsSweep(h->topo, h->bndInternal.data(), h->bs);
        return 0;
    case SMGPU_BOUNDARY_TABLES:
        if (!h->bndPending) return fail("smgpu_boundary_step: no set-up in progress (smgpu_boundary_begin, enabled)");
        return bndTables(h);
    case SMGPU_BOUNDARY_NORMALS_ACCUMULATE:   // SM.C:2219 (first calculateBoundaryPointNormals), local part
        if (!h->bndOn) return fail("smgpu_boundary_step: the tables step comes first");
        if (h->bndNormalsFromLayers) return 0;
        HIP_OK(hipMemsetAsync(h->st.acc, 0, sizeof(Accum), h->stream));
        if (h->bv.nB > 0) hipLaunchKernelGGL(k_bnd_normals, dim3(gridFor(h->bv.nB)), dim3(kBlock), 0, h->stream, m, h->st, h->bv);
        HIP_OK(hipStreamSynchronize(h->stream));
        HIP_OK(hipGetLastError());
        return 0;
    case SMGPU_BOUNDARY_NORMALS_FINISH:       // shared points, after the host has set the sums
        if (!h->bndOn) return fail("smgpu_boundary_step: the tables step comes first");
        if (h->bndNormalsFromLayers || !h->haloOn || !h->nShared) return 0;
        hipLaunchKernelGGL(k_bnd_normals_shared, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->st, h->bv, h->nShared, h->dSharedLocal);
        HIP_OK(hipStreamSynchronize(h->stream));
        return 0;
    default: return fail("smgpu_boundary_step: unknown step");
    }
}

int smgpu_boundary_shared(smgpu_handle* h, int32_t field, int32_t set, double* v) {
    if (!h || !v) return fail("null argument");
    if (!h->haloOn) return fail("smgpu_boundary_shared: halo not configured");
    HIP_OK(hipSetDevice(h->device));
    const std::vector<int>& sl = h->sharedLocalHost;
    const size_t n = sl.size();
    if (field == SMGPU_BOUNDARY_F_HOPS) {
        if (!h->bndPending) return fail("smgpu_boundary_shared: no set-up in progress");
        std::vector<int32_t>& hops = h->bs.hopsToSmoothingBoundary;
        for (size_t i = 0; i < n; ++i) { if (set) hops[(size_t)sl[i]] = (int32_t)v[i]; else v[i] = hops[(size_t)sl[i]]; }
        return 0;
    }
    if (field == SMGPU_BOUNDARY_F_NORMALS_COUNT) {   // 4 doubles: the local normal sum and the local boundary face count
        if (!h->bndOn) return fail("smgpu_boundary_shared: the tables step comes first");
        if (n == 0) return 0;
        const size_t W = (size_t)h->st.lStride;
        std::vector<double> rec(n * W, 0.0);
        if (!set) {
            // pack through the exchange kernel: the same values an iteration would send
            hipLaunchKernelGGL(k_halo_packL, dim3(gridFor(h->nShared)), dim3(kBlock), 0, h->stream, h->st,
                               PackLArgs{h->dSharedLocal, h->dOwnL, h->nShared, h->dSendOff, h->dSendSlots, h->sendL, h->bv.bfOff, h->bv.inner, h->bv.featOfBnd,
                                         h->bv.featSum, h->bv.featCnt});
            HIP_OK(hipMemcpyAsync(rec.data(), h->dOwnL, sizeof(double) * rec.size(), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
            for (size_t i = 0; i < n; ++i) { for (int c = 0; c < 3; ++c) v[4 * i + c] = rec[i * W + c]; v[4 * i + 3] = rec[i * W + 6]; }
        } else {
            for (size_t i = 0; i < n; ++i) { for (int c = 0; c < 3; ++c) rec[i * W + c] = v[4 * i + c]; rec[i * W + 6] = v[4 * i + 3]; }
            HIP_OK(hipMemcpyAsync(h->dCombL, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
        }
        return 0;
    }
    return fail("smgpu_boundary_shared: unknown field");
}

int smgpu_set_boundary_smoothing(smgpu_handle* h, const smgpu_boundary_desc* d, smgpu_boundary_info* info) {
    if (!h || !d) return fail("null argument");
    if (h->haloOn) return fail("smgpu_set_boundary_smoothing is the serial set-up; with a halo use smgpu_boundary_stats / begin / step / shared");
    double minEdge = 0.0, bb[6];
    if (smgpu_boundary_stats(h, &minEdge, bb)) return 1;
    smgpu_boundary_info bi{};
    if (smgpu_boundary_begin(h, d, minEdge, bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4] /* SM.C:1538 */, &bi)) return 1;
    if (info) *info = bi;
    if (!bi.enabled) return 0;
    for (int i = 0; i < 2; ++i) if (smgpu_boundary_step(h, SMGPU_BOUNDARY_HOPS_SWEEP)) return 1;   // SM.C:2218
    if (smgpu_boundary_step(h, SMGPU_BOUNDARY_TABLES)) return 1;
    return smgpu_boundary_step(h, SMGPU_BOUNDARY_NORMALS_ACCUMULATE);
}

int smgpu_get_boundary_classification(smgpu_handle* h, int32_t* isCornerPoint, int32_t* isFeatureEdgePoint) {
    if (!h || !isCornerPoint || !isFeatureEdgePoint) return fail("null argument");
    if ((int)h->bs.isCornerPointOut.size() != h->topo.nPoints) return fail("smgpu_get_boundary_classification: no boundary set-up has been made");
    std::copy(h->bs.isCornerPointOut.begin(), h->bs.isCornerPointOut.end(), isCornerPoint);
    std::copy(h->bs.isFeatureEdgePointOut.begin(), h->bs.isFeatureEdgePointOut.end(), isFeatureEdgePoint);
    return 0;
}

int smgpu_debug_edge_strings(int32_t nPoints, int32_t nEdges, const int32_t* edges, int32_t* strings, int32_t* nStrings) {
    if (nPoints < 0 || nEdges < 0 || (nEdges && (!edges || !strings))) return fail("bad argument");
    EdgeMeshHost em;
    em.pts.assign(3 * (size_t)nPoints, 0.0);
    em.edges.assign(edges, edges + 2 * (size_t)nEdges);
    for (int32_t v : em.edges) if (v < 0 || v >= nPoints) return fail("smgpu_debug_edge_strings: point id out of range");
    em.buildPointEdges();
    std::vector<int32_t> s;
    const int32_t n = edgeMeshStrings(em, s);
    std::copy(s.begin(), s.end(), strings);
    if (nStrings) *nStrings = n;
    return 0;
}

int smgpu_debug_find_line(smgpu_handle* h, int32_t n, const double* segments, double* hitPoints, int32_t* hit) {
    if (!h || !segments || !hitPoints || !hit || n < 0) return fail("bad argument");
    if (!h->bndOn) return fail("smgpu_debug_find_line: boundary point smoothing is not enabled");
    HIP_OK(hipSetDevice(h->device));
    double *dSeg = nullptr, *dOut = nullptr;
    int* dHit = nullptr;
    HIP_OK(hipMalloc((void**)&dSeg, sizeof(double) * 6 * (size_t)std::max(n, 1)));
    HIP_OK(hipMalloc((void**)&dOut, sizeof(double) * 3 * (size_t)std::max(n, 1)));
    HIP_OK(hipMalloc((void**)&dHit, sizeof(int) * (size_t)std::max(n, 1)));
    hipError_t e = hipMemcpy(dSeg, segments, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_bnd_find_line, dim3(gridFor(n)), dim3(kBlock), 0, h->stream, h->bv, n, dSeg, dOut, dHit);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(hitPoints, dOut, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hit, dHit, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(dSeg); (void)hipFree(dOut); (void)hipFree(dHit);
    if (e != hipSuccess) return fail(std::string("smgpu_debug_find_line: ") + hipGetErrorString(e));
    return 0;
}

// ---- debug / parity access -------------------------------------------------------------------
int smgpu_debug_propose(smgpu_handle* h) {
    if (!h) return fail("null handle");
    if (!h->prmSet) return fail("smgpu_set_params has not been called");
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipMemsetAsync(h->st.acc, 0, sizeof(Accum), h->stream));
    h->writeFaces = true;
    if (runBndPre(h)) return 1;
    const int rcg = runGeometry(h, nullptr, 0, false, true);
    h->writeFaces = false;
    if (rcg) return 1;
    // the debug fields (edge / point angles) are only complete without the filters; SMGPU_DEBUG_FILTERED=1
    // keeps them on so tests can compare the decisions of the filtered path
    h->exactAll = envInt("SMGPU_DEBUG_FILTERED", 0) == 0;
    const int rcp = runProposalAndConstraints(h);
    h->exactAll = false;
    if (rcp) return 1;
    return checkDeviceError(h);
}

int smgpu_debug_get_field(smgpu_handle* h, const char* name, double* out, int64_t* n) {
    if (!h || !name || !n) return fail("null argument");
    HIP_OK(hipSetDevice(h->device));
    const std::string s(name);
    const State& st = h->st;
    const int64_t P = h->mv.nPoints, C = h->mv.nCells, F = h->mv.nFaces, E = h->mv.nEdges;
    const double* dsrc = nullptr;
    const uint8_t* bsrc = nullptr;
    int64_t cnt = 0;
    if (s == "cellCentres") { dsrc = st.cellCtr; cnt = 3 * C; }
    else if (s == "faceCentres") { dsrc = st.fCtr; cnt = 3 * F; }
    else if (s == "faceAreas") { dsrc = st.fArea; cnt = 3 * F; }
    else if (s == "newPoints") { dsrc = st.prop; cnt = 3 * P; }
    else if (s == "points") { dsrc = st.ptsCur; cnt = 3 * P; }
    else if (s == "edgeMinAngle") { dsrc = st.edgeMin; cnt = E; }
    else if (s == "edgeMaxAngle") { dsrc = st.edgeMax; cnt = E; }
    else if (s == "pointMinAngle") { dsrc = st.ptMin; cnt = P; }
    else if (s == "pointMaxAngle") { dsrc = st.ptMax; cnt = P; }
    else if (s == "isFrozenPoint") { bsrc = st.frozen; cnt = P; }
    else if (s == "faActive") { bsrc = st.faActive; cnt = P; }
    else if (s == "layerNormals" && (h->layersOn || h->bndOn)) { dsrc = st.layerNormal; cnt = 3 * P; }
    else if ((s == "layerHops" || s == "layerOuterMap") && h->layersOn) {
        *n = P;
        if (out) { const std::vector<int32_t>& v = (s == "layerHops") ? h->layerHopsHost : h->layerMapHost; for (int64_t i = 0; i < P; ++i) out[i] = v[(size_t)i]; }
        return 0;
    }
    else return fail("unknown field " + s);
    *n = cnt;
    if (!out) return 0;
    if (dsrc) {
        HIP_OK(hipMemcpyAsync(out, dsrc, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
    } else {
        std::vector<uint8_t> tmp((size_t)cnt);
        HIP_OK(hipMemcpyAsync(tmp.data(), bsrc, (size_t)cnt, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        const bool tagged = (bsrc == st.faActive);   // marks carry the iteration's tag
        for (int64_t i = 0; i < cnt; ++i) out[i] = tagged ? (tmp[(size_t)i] == st.faGen ? 1.0 : 0.0) : (double)tmp[(size_t)i];
    }
    return 0;
}

static int topoGet(const Topology& t, const char* kind, int32_t* offsets, int32_t* values, int64_t* nnz) {
    const std::string s(kind);
    const std::vector<int32_t>* off = nullptr;
    const std::vector<int32_t>* val = nullptr;
    if (s == "pointCells") { off = &t.pointCells.off; val = &t.pointCells.val; }
    else if (s == "pointPoints") { off = &t.pointEdges.off; val = &t.pointPoints; }
    else if (s == "pointEdges") { off = &t.pointEdges.off; val = &t.pointEdges.val; }
    else if (s == "pointFaces") { off = &t.pointFaces.off; val = &t.pointFaces.val; }
    else if (s == "pointFacePrev") { off = &t.pointFaces.off; val = &t.pfPrev; }
    else if (s == "pointFaceNext") { off = &t.pointFaces.off; val = &t.pfNext; }
    else if (s == "edgeFaces") { off = &t.edgeFaces.off; val = &t.edgeFaces.val; }
    else if (s == "edgeCells") { off = &t.edgeCells.off; val = &t.edgeCells.val; }
    else if (s == "cellFacesGeom") { off = &t.cellFacesGeom.off; val = &t.cellFacesGeom.val; }
    else if (s == "edges") { val = &t.edges; }
    else return fail("unknown addressing " + s);
    *nnz = (int64_t)val->size();
    if (offsets && off) std::memcpy(offsets, off->data(), sizeof(int32_t) * off->size());
    if (values) std::memcpy(values, val->data(), sizeof(int32_t) * val->size());
    return 0;
}

// FNV-1a checksum of every array of the addressing, in a fixed order (SMGPU_TOPO_CHECKSUMS entries): the device build
// (topology_dev.hip, what an engine holds) against the host build (smgpu_topology_create) in tests/test_gpu_topology.py
static uint64_t fnv1a(const void* p, size_t n, uint64_t hh = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { hh ^= b[i]; hh *= 1099511628211ull; }
    return hh;
}
static void topoChecksums(const Topology& t, uint64_t* out) {
    int k = 0;
    const int32_t sizes[9] = {t.nPoints, t.nCells, t.nFaces, t.nInternalFaces, t.nEdges, t.maxFaceSize, t.maxEdgeFaces, t.maxPointCells, t.maxPointPoints};
    out[k++] = fnv1a(sizes, sizeof(sizes));
    auto add = [&](const auto& v) { out[k++] = fnv1a(v.data(), v.size() * sizeof(v[0])) ^ (uint64_t)v.size(); };
    add(t.facePoints.off); add(t.facePoints.val); add(t.owner); add(t.neighbour);
    add(t.cellFacesGeom.off); add(t.cellFacesGeom.val);
    add(t.pointFaces.off); add(t.pointFaces.val); add(t.pfPrev); add(t.pfNext); add(t.pfPrevSlot); add(t.pfNextSlot);
    add(t.pointCells.off); add(t.pointCells.val);
    add(t.edges); add(t.pointEdges.off); add(t.pointEdges.val); add(t.pointPoints);
    add(t.edgeFaces.off); add(t.edgeFaces.val); add(t.edgeCells.off); add(t.edgeCells.val); add(t.ecFace0); add(t.ecFace1);
    add(t.ringFace); add(t.ringCell); add(t.edgeRingOk);
    while (k < SMGPU_TOPO_CHECKSUMS) out[k++] = 0;
}
// checksums of the geometry tile tables as the kernels read them (downloaded from the device), in a fixed order: the device
// build of the tables (tiles_dev.hip) against the host build (SMGPU_DEVICE_TILES=0) in tests/test_gpu_topology.py
int smgpu_debug_tile_checksums(smgpu_handle* h, uint64_t* out) {
    if (!h || !out) return fail("null argument");
    for (int i = 0; i < SMGPU_TOPO_CHECKSUMS; ++i) out[i] = 0;
    if (!h->useTiles) return 0;
    HIP_OK(hipSetDevice(h->device));
    const GeomTiles& gt = h->gt;
    const int nT = gt.nTiles;
    int k = 0;
    auto host = [&](const auto& v) { out[k++] = fnv1a(v.data(), v.size() * sizeof(v[0])) ^ (uint64_t)v.size(); };
    auto dev = [&](const void* p, size_t bytes) -> int {
        std::vector<unsigned char> buf(bytes);
        if (bytes) HIP_OK(hipMemcpy(buf.data(), p, bytes, hipMemcpyDeviceToHost));
        out[k++] = fnv1a(buf.data(), bytes) ^ (uint64_t)bytes;
        return 0;
    };
    host(gt.cellBeg); host(gt.tpOff); host(gt.tfOff); host(gt.fvBase); host(gt.fvWidth); host(gt.cfBase); host(gt.cfWidth); host(gt.tileFlags);
    const int32_t mx[2] = {gt.maxPoints, gt.maxFaces};
    out[k++] = fnv1a(mx, sizeof(mx));
    const size_t nTp = (size_t)gt.tpOff.back(), nTf = (size_t)gt.tfOff.back();
    const size_t nFv = nT ? (size_t)gt.fvBase[(size_t)nT - 1] + (size_t)(gt.tfOff[(size_t)nT] - gt.tfOff[(size_t)nT - 1]) * gt.fvWidth[(size_t)nT - 1] : 0;
    const size_t nCf = nT ? (size_t)gt.cfBase[(size_t)nT - 1] + (size_t)gt.cfWidth[(size_t)nT - 1] * (size_t)gt.threads : 0;
    if (dev(h->gv.cellOrder, (size_t)h->mv.nCells * 4) || dev(h->gv.tpIds, nTp * 4) || dev(h->gv.tfIds, nTf * 4) || dev(h->gv.faceVerts, nFv * 2) ||
        dev(h->gv.cellFaces, nCf * 2) || dev(h->gv.meta, (size_t)nT * kGeomMetaInts * 4)) return 1;
    {      // the smoothing tiles
        const SmoothTiles& st = h->stl;
        const int nS = st.nTiles;
        host(st.ptBeg); host(st.tcOff); host(st.tnOff); host(st.pcBase); host(st.pcWidth); host(st.ppBase); host(st.ppWidth); host(st.pfBase); host(st.pfWidth);
        const int32_t mxS[2] = {st.maxCells, st.maxPoints};
        out[k++] = fnv1a(mxS, sizeof(mxS));
        const size_t nPc = nS ? (size_t)st.pcBase[(size_t)nS - 1] + (size_t)st.pcWidth[(size_t)nS - 1] * (size_t)st.threads : 0;
        const size_t nPp = nS ? (size_t)st.ppBase[(size_t)nS - 1] + (size_t)st.ppWidth[(size_t)nS - 1] * (size_t)st.threads : 0;
        const size_t nPfE = nS ? (size_t)st.pfBase[(size_t)nS - 1] + (size_t)st.pfWidth[(size_t)nS - 1] * (size_t)st.threads : 0;
        if (dev(h->sv.ptOrder, (size_t)h->mv.nPoints * 4) || dev(h->sv.tcIds, (size_t)st.tcOff.back() * 4) || dev(h->sv.tnIds, (size_t)st.tnOff.back() * 4) ||
            dev(h->sv.selfLoc, (size_t)h->mv.nPoints * 2) || dev(h->sv.pcEll, nPc * 2) || dev(h->sv.ppEll, nPp * 2) || dev(h->sv.pairEll, nPp * 2) || dev(h->sv.pfEll, nPfE * 2) ||
            dev(h->sv.meta, (size_t)nS * kSmoothMetaInts * 4)) return 1;
    }
    if (h->useFilter && h->ev.meta) {      // the edge tiles
        const EdgeTiles& et = h->etl;
        const int nE = et.nTiles;
        host(et.tpOff); host(et.tfOff); host(et.tcOff); host(et.efBase); host(et.ecBase); host(et.efWidth); host(et.ecWidth);
        const int32_t mxE[3] = {et.maxPoints, et.maxFaces, et.maxCells};
        out[k++] = fnv1a(mxE, sizeof(mxE));
        const size_t nEf = nE ? (size_t)et.efBase[(size_t)nE - 1] + (size_t)et.efWidth[(size_t)nE - 1] * (size_t)et.threads : 0;
        const size_t nEc = nE ? (size_t)et.ecBase[(size_t)nE - 1] + (size_t)et.ecWidth[(size_t)nE - 1] * (size_t)et.threads : 0;
        if (dev(h->ev.order, (size_t)h->mv.nEdges * 4) || dev(h->ev.edgeBeg, ((size_t)nE + 1) * 4) || dev(h->ev.tpIds, (size_t)et.tpOff.back() * 4) ||
            dev(h->ev.tfIds, (size_t)et.tfOff.back() * 4) || dev(h->ev.tcIds, (size_t)et.tcOff.back() * 4) || dev(h->ev.epLoc, (size_t)h->mv.nEdges * 4) ||
            dev(h->ev.efEll, nEf * 2) || dev(h->ev.ecEll, nEc * 2) || dev(h->ev.meta, (size_t)nE * kEdgeMetaInts * 4)) return 1;
    }
    return 0;
}

int smgpu_debug_addressing_checksums(smgpu_handle* h, uint64_t* out) {
    if (!h || !out) return fail("null argument");
    if (ensureHostLists(h, 3)) return 1;
    topoChecksums(h->topo, out);
    return 0;
}

int smgpu_debug_get_addressing(smgpu_handle* h, const char* kind, int32_t* offsets, int32_t* values, int64_t* nnz) {
    if (!h || !kind || !nnz) return fail("null argument");
    if (ensureHostLists(h, 1)) return 1;
    return topoGet(h->topo, kind, offsets, values, nnz);
}

struct smgpu_topology { Topology t; };

int smgpu_topology_create(const smgpu_mesh_desc* d, smgpu_topology** out) {
    if (!d || !out) return fail("null argument");
    smgpu_topology* t = new smgpu_topology();
    const std::string terr = t->t.build(d->nPoints, d->nCells, d->nFaces, d->nInternalFaces, d->faceOffsets, d->facePoints,
                                        d->owner, d->neighbour);
    if (!terr.empty()) { delete t; *out = nullptr; return fail("smgpu_topology_create: " + terr); }
    *out = t;
    return 0;
}
int smgpu_topology_get(smgpu_topology* t, const char* kind, int32_t* offsets, int32_t* values, int64_t* nnz) {
    if (!t || !kind || !nnz) return fail("null argument");
    return topoGet(t->t, kind, offsets, values, nnz);
}
int smgpu_topology_checksums(smgpu_topology* t, uint64_t* out) {
    if (!t || !out) return fail("null argument");
    topoChecksums(t->t, out);
    return 0;
}
int smgpu_topology_num_edges(smgpu_topology* t, int32_t* nEdges) {
    if (!t || !nEdges) return fail("null argument");
    *nEdges = t->t.nEdges;
    return 0;
}
int smgpu_topology_destroy(smgpu_topology* t) { delete t; return 0; }

}  // extern "C"
