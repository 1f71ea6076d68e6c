/* smgpu.h -- C-ABI of the MI355X (gfx950) smoothing-iteration engine.
 *
 * Drop-in boundary for the hot path of tkeskita/smoothMesh: the body of the smoothing loop
 * src/smoothMesh.C:2257-2437 (cell-centre recompute, centroidalSmoothing :96-166,
 * aspectRatioSmoothing :548-593, constrainMaxStepLength :684-754, restrictEdgeShortening
 * :602-652, restrictMinEdgeAngleDecrease :900-930, restrictFaceAngleDeterioration :1320-1437,
 * restore/count :2384-2392, calculateResidual :1546-1570, mesh.movePoints :2399).
 *
 * The reference has no FFI: those are free functions over OpenFOAM types (const fvMesh&,
 * pointField&, boolList&) called from main().  A host (the bundled `smoothMesh` front-end in
 * smoothmesh_amd/csrc/host, or an OpenFOAM-linked main) hands over what a
 * polyMesh directory contains -- points, faces, owner, neighbour, point masks -- as plain
 * arrays; all other addressing is derived inside the library.
 *
 * Conventions: every call returns 0 on success, non-zero on error (message through
 * smgpu_last_error); the reference aborts the process on FatalError, the host does the same.
 * Host arrays are borrowed for the duration of the call only.  A handle owns one HIP device
 * stream and all its device memory; it is not thread-safe.  label = int32, scalar = f64,
 * point = 3 x f64 (AoS), boolList = 1 byte per element.
 */
#ifndef SMGPU_H
#define SMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smgpu_handle smgpu_handle;

/* What createMesh.H (SM.C:1814-1818) + findInternalMeshPoints (SM.C:40-91) +
 * classifyBoundaryPoints (BPS.C:269-441) give the loop. */
typedef struct smgpu_mesh_desc {
    int32_t nPoints, nCells, nFaces, nInternalFaces;
    const double* points;           /* [3*nPoints]            mesh.points()                  */
    const int32_t* faceOffsets;     /* [nFaces+1]             CSR of mesh.faces()            */
    const int32_t* facePoints;      /* [faceOffsets[nFaces]]                                  */
    const int32_t* owner;           /* [nFaces]               polyMesh::faceOwner()          */
    const int32_t* neighbour;       /* [nInternalFaces]       polyMesh::faceNeighbour()      */
    const uint8_t* isInternalPoint;          /* [nPoints]  SM.C:1978-1979                    */
    const uint8_t* isSmoothingSurfacePoint;  /* [nPoints]  BPS.C:404-412; NULL = all false   */
    int32_t device;                 /* HIP device ordinal                                     */
    void* stream;                   /* hipStream_t to run on (with useCallerStream)           */
    int32_t useCallerStream;        /* 0: library creates its own stream and ignores `stream`;
                                       1: run on `stream` as given -- NULL then means the HIP null
                                       stream (what torch.cuda.current_stream() is by default)   */
} smgpu_mesh_desc;

/* Loop parameters, SM.C:1861-1890 (names as the command-line options). */
typedef struct smgpu_params {
    double maxStepLength;
    double relStepFrac;
    double minEdgeLength;
    int32_t totalMinFreeze;
    int32_t edgeAngleConstraint;
    int32_t faceAngleConstraint;
    double minAngle;   /* degrees */
    double maxAngle;   /* degrees */
} smgpu_params;

/* One line of "Smoothing iteration=N nFrozenPoints=M residual=R" (SM.C:2396). */
typedef struct smgpu_iter_stats {
    double residual;
    int32_t nFrozenPoints;
    /* Near-tie census of this iteration (0 in a normal run): comparisons of an angle with minAngle / maxAngle or with the point's
     * current angle (SM.C:923, 1367, 1391-1394, 1421-1424) whose two sides were 1 .. 4 ulp apart.  The engine's acos and the
     * reference's (glibc) may differ in the last bit, so only such a comparison could be decided differently by the reference. */
    int32_t nNearTies;
} smgpu_iter_stats;

/* Mesh sizes derived by the library (for byte accounting and logs). */
typedef struct smgpu_sizes {
    int64_t nPoints, nCells, nFaces, nInternalFaces, nEdges;
    int64_t nnzFacePoints, nnzPointCells, nnzPointPoints, nnzPointFaces, nnzEdgeFaces, nnzEdgeCells, nnzCellFaces;
    int64_t deviceBytes;
} smgpu_sizes;

/* Per-kernel accumulated device time (hipEvent, milliseconds) and launch counts. */
#define SMGPU_MAX_KERNELS 24
typedef struct smgpu_counters {
    int32_t nKernels;
    const char* name[SMGPU_MAX_KERNELS];
    double ms[SMGPU_MAX_KERNELS];
    int64_t launches[SMGPU_MAX_KERNELS];
    int64_t algoBytesPerLaunch[SMGPU_MAX_KERNELS];
    int64_t algoF64OpsPerLaunch[SMGPU_MAX_KERNELS];   /* FP64 VALU instructions (per element, not per wave) the
                                                         reference's arithmetic needs, sqrt = 22 and div = 11 as the
                                                         compiler expands them; 0 = not counted for this kernel    */
} smgpu_counters;

const char* smgpu_last_error(void);
const char* smgpu_version(void);

/* Build addressing on the host, upload everything.  Replaces the setup SM.C:1978-2021. */
int smgpu_create(const smgpu_mesh_desc* desc, smgpu_handle** out);
int smgpu_destroy(smgpu_handle* h);

int smgpu_get_sizes(smgpu_handle* h, smgpu_sizes* out);
/* getMeshStats (SM.C:1478-1541): min / max edge length of the current coordinates. */
int smgpu_mesh_stats(smgpu_handle* h, double* minEdgeLength, double* maxEdgeLength);
int smgpu_set_params(smgpu_handle* h, const smgpu_params* p);
/* Which OpenFOAM's primitiveMesh geometry the cell centres follow (row a3 of the scope table; the reference is built against
 * either line, Allwmake:47 / README.md:30, and inherits its makeFaceCentresAndAreas / makeCellCentresAndVols):
 * COM = OpenFOAM.com v2312-v2506 (default; fan triangles of a face weighted by their area magnitude), ORG = OpenFOAM.org 12
 * (weighted by the area projected on the face normal; pyramid volumes clamped at vSmall).  The environment variable
 * SMGPU_FOAM_VARIANT=org selects ORG at smgpu_create.  Call before iterating. */
enum { SMGPU_FOAM_COM = 0, SMGPU_FOAM_ORG = 1 };
int smgpu_set_foam_variant(smgpu_handle* h, int32_t variant);
/* Multi-rank runs: which syncTools::syncPointList the shared-point combines of minMagSqrEqOp / maxMagSqrEqOp follow (SM.C:402-469
 * with isCloserPoint :246-272, OBB.C:359-365, :490-496).  MASTER (default) = globalMeshData::syncData, the form of every OpenFOAM
 * the reference builds against (Allwmake:47): the values of a point's sharers are folded ONCE, starting from the lowest rank's
 * value, in ascending rank order, and every sharer receives that result -- an exact tie hands the lower rank's vector to all, so a
 * rank can receive another rank's equal-length vector (the case isCloserPoint exists for) and a run cut through an exactly graded
 * mesh reproduces the serial aspect-ratio blend.  OWN = every sharer folds the others' values onto its own (a tie keeps the own
 * vector; the sharers may end with different vectors): the model of rounds 1-3 of this library, kept as the A/B.  The
 * environment variable SMGPU_SYNC_VARIANT=own selects OWN at smgpu_create.  The host-side combines of the step-wise set-ups
 * (smgpu_layers_shared / smgpu_boundary_shared) are the host's: smoothmesh_amd/halo.py and the smoothMesh front-end follow the
 * same switch.  Call before iterating. */
enum { SMGPU_SYNC_MASTER = 0, SMGPU_SYNC_OWN = 1 };
int smgpu_set_sync_variant(smgpu_handle* h, int32_t variant);

/* The loop SM.C:2257-2437 on one rank: up to nIters iterations, stops after the first iteration
 * whose residual < relTol (SM.C:2401).  stats (host, [nIters], may be NULL) receives one entry
 * per iteration done.  No host synchronisation happens inside the loop -- with one bounded exception while the face-angle
 * constraint is on: the form of the freeze-walk replay follows the number of points outside the good angle range the GPU has
 * published (updateWalkMode), and to bound how stale that number can be the host waits, every 8th iteration, for the iteration
 * it enqueued 8 iterations earlier (at most ~16 iterations are queued ahead; the GPU never idles).  The same holds for the
 * step-wise loop (smgpu_iter_begin). */
int smgpu_iterate(smgpu_handle* h, int32_t nIters, double relTol, smgpu_iter_stats* stats, int32_t* nDone);

/* smgpu_get_points also reports an error word a kernel raised since the last check (the step-wise loop below never
 * synchronises on its own): a non-zero return with smgpu_last_error() set. */
int smgpu_get_points(smgpu_handle* h, double* outPoints /* [3*nPoints] */);
/* Waits for the engine's stream and reports any error word a kernel has raised since the last check (a grid barrier or a
 * peer-store wait that timed out, a point without two usable neighbours SM.C:354-362, a projection failure BPS.C:932-938 ...).
 * The step-wise loop never synchronises on its own; hosts call this at the end of every chunk of iterations (smgpu_get_points
 * does the same check). */
int smgpu_check_error(smgpu_handle* h);
/* Near-tie census since smgpu_create (see smgpu_iter_stats::nNearTies; also counted in the step-wise multi-rank loop):
 * out = {total, edge-angle test SM.C:923, good-range test SM.C:1367, walk verdicts SM.C:1391-1394 / 1421-1424}.  The front-ends
 * print a warning when the total is not zero.  Window: SMGPU_NEARTIE_ULPS (default 4). */
int smgpu_get_near_ties(smgpu_handle* h, int64_t out[4]);
int smgpu_set_points(smgpu_handle* h, const double* points /* [3*nPoints] */);

/* Timing: when enabled every kernel launch is bracketed by hipEvents on the handle's stream. */
int smgpu_enable_timing(smgpu_handle* h, int32_t on);
int smgpu_get_counters(smgpu_handle* h, smgpu_counters* out);
int smgpu_reset_counters(smgpu_handle* h);

/* ---- multi-rank (domain decomposition, one handle per rank/GPU) --------------------------
 * Replaces syncTools::syncPointList at SM.C:134,142,402,429,455,472 (one fused exchange "A")
 * and SM.C:2374 (exchange "F").  The library packs/unpacks; the host moves the buffers
 * (RCCL all_to_all over xGMI) between the calls.  All buffers are DEVICE pointers owned by the
 * caller.  Record of exchange A = 13 doubles per slot {sum xyz, r1 xyz, r2 xyz, r3 xyz,
 * (int32 count, int32 hasCommonCell)}; record of exchange F = 1 int32 per slot. */
#define SMGPU_HALO_A_DOUBLES 13
typedef struct smgpu_halo_desc {
    int32_t nShared;              /* points of this rank shared with >= 1 other rank          */
    const int32_t* sharedLocal;   /* [nShared] local point ids                                */
    int32_t nSend;                /* send slots (concatenated by destination rank, ascending) */
    const int32_t* sendShared;    /* [nSend]   index into sharedLocal of each send slot       */
    int32_t nRecv;                /* recv slots (concatenated by source rank, ascending)      */
    const int32_t* combOffsets;   /* [nShared+1] CSR over sharers of each shared point        */
    const int32_t* combSlots;     /* recv slot of each sharer, ascending rank; -1 = this rank */
    void* sendA; void* recvA;     /* device, nSend / nRecv records of 13 doubles              */
    void* sendF; void* recvF;     /* device, nSend / nRecv int32                              */
    void* localStats;             /* device, 2 doubles {residual, nFrozenPoints} per iteration */
    void* sendL; void* recvL;     /* device, nSend / nRecv records sized for SMGPU_HALO_L_DOUBLES doubles; only used with the
                                     boundary layer treatment / boundary point smoothing (may be NULL otherwise),
                                     exchanged together with sendA / recvA (smgpu_halo_l_doubles per slot in use)  */
    int32_t useExchangeStream;    /* 0: the host enqueues its exchanges on the engine's stream (in order).       */
    void* exchangeStream;         /* 1: the host enqueues them on THIS hipStream_t (NULL = the null stream); the
                                     engine orders its own stream against it with events inside
                                     smgpu_iter_begin/mid/end, so the exchange runs next to the kernels of
                                     smgpu_iter_interior / smgpu_iter_ahead and the compute queue does not idle */
} smgpu_halo_desc;
int smgpu_halo_configure(smgpu_handle* h, const smgpu_halo_desc* d);
/* ---- peer-store transport: the ranks of one node move the shared-point records THEMSELVES -------------------------------
 * Instead of the host enqueuing an exchange between smgpu_iter_begin / mid / end (MPI, RCCL send / recv groups), the kernels
 * that pack exchange A / L / F store every record straight into the peer's receive slot over xGMI (the peers' receive
 * buffers are mapped into this process: hipIpc) and raise a flag word at the peer; the first kernel that consumes the records
 * waits for its peers' flags.  pack = send, no collective kernel, no host call.  The host
 *   1. allocates recvA / recvL / recvF and one flag array (uint32[2 * 64], zero) with smgpu_push_alloc and passes the
 *      receive buffers to smgpu_halo_configure as usual (useExchangeStream = 0);
 *   2. hands the 64-byte handles to its peers (any channel) and maps theirs with smgpu_push_open;
 *   3. calls smgpu_halo_set_push on every rank, then iterates WITHOUT moving anything between the smgpu_iter_* calls.
 * peers = the ranks this rank shares points with, ascending (the order of the send slot groups); remoteBase[o] = first slot,
 * in peer o's receive numbering, of this rank's records; myIndexAtPeer[o] = this rank's position among peer o's peers.
 * All ranks must use the same transport.  A record that does not arrive within two seconds raises an error at the next
 * host read-back (smgpu_last_error).  d = NULL switches back to host-driven exchanges. */
typedef struct smgpu_push_desc {
    int32_t nPeers;
    const int32_t* peerCount;       /* [nPeers] slots per peer (sum = nSend)                              */
    const int32_t* remoteBase;      /* [nPeers]                                                           */
    const int32_t* myIndexAtPeer;   /* [nPeers]                                                           */
    void* const* peerRecvA;         /* [nPeers] the peers' buffers, mapped                                */
    void* const* peerRecvL;         /* [nPeers] (entries may be NULL while neither layers nor boundary smoothing is on) */
    void* const* peerRecvF;         /* [nPeers]                                                           */
    void* const* peerFlags;         /* [nPeers] the peers' flag arrays, mapped                            */
    void* localFlags;               /* this rank's flag array                                             */
} smgpu_push_desc;
int smgpu_halo_set_push(smgpu_handle* h, const smgpu_push_desc* d);
int smgpu_push_alloc(int32_t device, size_t bytes, void** ptr, void* ipcHandle64);   /* uncached device memory + its IPC handle */
int smgpu_push_open(int32_t device, const void* ipcHandle64, void** ptr);            /* map a peer's allocation               */
int smgpu_push_close(void* ptr);
int smgpu_push_free(void* ptr);

/* change useExchangeStream / exchangeStream of a configured halo (e.g. to time both arrangements on the target) */
int smgpu_halo_set_exchange_stream(smgpu_handle* h, int32_t useExchangeStream, void* exchangeStream);
/* the hipStream_t the engine launches on (its own, or the caller's when useCallerStream was set) */
int smgpu_get_stream(smgpu_handle* h, void** stream);
/* optional: smgpu_iter_end additionally writes the iteration's {residual, nFrozenPoints} to record (n mod capacity) of
 * this device array of 2*capacity doubles, n = number of smgpu_iter_end calls since this call -- a run that cannot
 * stop early (relTol <= 0) then needs no per-iteration host action on localStats.  history = NULL switches it off. */
int smgpu_halo_set_stats_history(smgpu_handle* h, void* history, int32_t capacity);
int smgpu_iter_begin(smgpu_handle* h);   /* geometry + local partial sums / closest points -> sendA */
int smgpu_iter_interior(smgpu_handle* h);/* optional, between begin and mid: everything that does not need
                                            recvA (points away from the shared ones) -- lets the host overlap
                                            exchange A with compute                                         */
int smgpu_iter_mid(smgpu_handle* h);     /* combine recvA, proposal, constraints -> sendF           */
int smgpu_iter_ahead(smgpu_handle* h);   /* optional, between mid and end: next iteration's geometry away from
                                            the shared points -- overlaps exchange F (constraints off)      */
int smgpu_iter_end(smgpu_handle* h);     /* or recvF, restore, residual -> localStats; movePoints   */

/* ---- optional boundary layer treatment (prismatic layers on selected patches) -------------------------------
 * Replaces SM.C:2186-2221 (set-up: point classification BPS.C:296-340,397-403; calculatePointHopsToBoundary,
 * calculateBoundaryPointNormals, propagateOuterNeighInfo, OBB.C = src/orthogonalBoundaryBlending.C) and, inside
 * every later iteration, SM.C:2266 + 2283-2305 (updateNeighCoords, blendWithOrthogonalPoints, second step clamp).
 * Patches as in polyMesh/boundary: face ranges in file order.  Call after smgpu_create (the set-up uses the
 * coordinates the engine holds) and before iterating.  With a halo (smgpu_halo_configure) use the step-wise form below.
 * *enabled = the reference's doLayerTreatment (a layer patch is selected and layerMaxBlendingFraction > SMALL). */
typedef struct smgpu_layer_desc {
    int32_t nPatches;
    const int32_t* patchStart;     /* [nPatches] first face                                    */
    const int32_t* patchSize;      /* [nPatches] number of faces                               */
    const uint8_t* patchKind;      /* [nPatches] 0 ordinary, 1 processor, 2 empty (OBB.C:156-159) */
    const uint8_t* isLayerPatch;   /* [nPatches] selected by -layerPatches (SM.C:1823)          */
    double layerMaxBlendingFraction;   /* SM.C:1892, default 0.3                                */
    double layerEdgeLength;            /* SM.C:1895, default minEdgeLength                      */
    double layerExpansionRatio;        /* SM.C:1898, default 1.3                                */
    int32_t minLayers, maxLayers;      /* SM.C:1901-1905, defaults 1 and 4                      */
} smgpu_layer_desc;
int smgpu_set_layers(smgpu_handle* h, const smgpu_layer_desc* d, int32_t* enabled);

/* The same set-up in steps, for runs with a halo (-parallel): between the steps the host performs the reference's
 * syncTools::syncPointList calls over the shared points (values in the order of smgpu_halo_desc.sharedLocal, exchanged
 * and combined by the host like exchange A).  Sequence (SM.C:2215-2221):
 *   smgpu_layers_begin                                         -> *maxIter = maxLayers + 1
 *   maxIter x { step HOPS_SWEEP;  get HOPS, combine with max over the sharers (OBB.C:124-130), set HOPS }
 *   step NORMALS_ACCUMULATE;  get NORMALS_COUNT, sum over the sharers in ascending rank order (OBB.C:184-198), set
 *   step NORMALS_FINISH
 *   for iter = 1..maxIter { step PROPAGATE_SWEEP(iter);  get NORMALS, every sharer folds the others' values onto its own
 *                           in ascending rank order keeping the larger magnitude, ties keep (OBB.C:359-365), set }
 *   step FINISH
 * Every later iteration additionally exchanges 6 doubles per shared point (smgpu_halo_desc.sendL / recvL: the local
 * normal and the outer neighbour's coordinates; plusEq OBB.C:184-198 and minMagSqr OBB.C:490-496) next to exchange A. */
enum { SMGPU_LAYERS_HOPS_SWEEP = 0, SMGPU_LAYERS_NORMALS_ACCUMULATE = 1, SMGPU_LAYERS_NORMALS_FINISH = 2,
       SMGPU_LAYERS_PROPAGATE_SWEEP = 3, SMGPU_LAYERS_FINISH = 4 };
enum { SMGPU_LAYERS_F_HOPS = 0,            /* 1 double per shared point (the hop count or -1)        */
       SMGPU_LAYERS_F_NORMALS_COUNT = 1,   /* 4 doubles: normal, number of boundary faces            */
       SMGPU_LAYERS_F_NORMALS = 2 };       /* 3 doubles: normal                                      */
/* One record per shared point: [0:3] local normal, [3:6] outer neighbour coordinates (layers), [6] local number of boundary
 * faces, [7:10] inner neighbour coordinates, [10:13] local feature edge projection sum, [13] its count (boundary point
 * smoothing; zero / UNDEF when that is off).  Exchanged whenever the layer treatment or the boundary point smoothing is on. */
#define SMGPU_HALO_L_DOUBLES 14
/* The record in use is shorter while only the layer treatment needs it: the first SMGPU_HALO_L_LAYERS doubles, records packed
 * back to back from the start of sendL / recvL (which are sized for SMGPU_HALO_L_DOUBLES per slot).  smgpu_halo_l_doubles tells
 * the host how many doubles per slot to exchange; it changes when the boundary point smoothing is set up. */
#define SMGPU_HALO_L_LAYERS 6
int smgpu_halo_l_doubles(smgpu_handle* h, int32_t* doublesPerSlot);
int smgpu_layers_begin(smgpu_handle* h, const smgpu_layer_desc* d, int32_t* enabled, int32_t* maxIter);
int smgpu_layers_step(smgpu_handle* h, int32_t step, int32_t arg);
int smgpu_layers_shared(smgpu_handle* h, int32_t field, int32_t set, double* values);

/* ---- optional boundary point smoothing (projection of boundary points to feature edges and target surfaces) --------
 * Replaces, for this feature, the set-up SM.C:2080-2253 (edge mesh sanity checks BPS.C:20-79, target edge strings
 * BPS.C:446-587, classifyBoundaryPoints BPS.C:269-441, hop counts to the smoothing patches OBB.C:52-133, inner neighbour
 * map OBB.C:396-459, target strings of the feature edge points SM.C:2234-2249) and, inside every later iteration,
 * SM.C:2266 (calculateBoundaryPointNormals OBB.C:141-233), centroidalSmoothing of the boundary points too (SM.C:116),
 * SM.C:2307-2357 (projectBoundaryPointsToEdgesAndSurfaces BPS.C:843-945, projectPrismaticInternalPointsToSurfaces
 * OBB.C:573-631, third step clamp)  (BPS.C = src/boundaryPointSmoothing.C, OBB.C = src/orthogonalBoundaryBlending.C).
 * Inputs are the contents of constant/geometry/{initEdges,targetEdges,targetSurfaces}.obj (SM.C:1924-1926) as flat
 * arrays and, optionally, the isCornerPoint / isFeatureEdgePoint lists a previous run wrote (SM.C:2039-2077).
 * OpenFOAM's octree line query is replaced by a bounding volume hierarchy; semantics in csrc/kernels_boundary.hpp.
 * Call after smgpu_create and, when both are used, after smgpu_set_layers; before iterating.  With a halo use the
 * step-wise form below.
 * info->enabled = the reference's doBoundarySmoothing (SM.C:2080-2093). */
typedef struct smgpu_boundary_desc {
    int32_t nPatches;
    const int32_t* patchStart;         /* [nPatches] first face                                              */
    const int32_t* patchSize;          /* [nPatches] number of faces                                         */
    const uint8_t* patchKind;          /* [nPatches] 0 ordinary, 1 processor, 2 empty                        */
    const uint8_t* isSmoothingPatch;   /* [nPatches] selected by -smoothingPatches (default all, SM.C:1837)  */
    int32_t nInitEdgePoints;   const double* initEdgePoints;     /* [3 n] initEdges.obj                      */
    int32_t nInitEdges;        const int32_t* initEdges;         /* [2 n]                                    */
    int32_t nTargetEdgePoints; const double* targetEdgePoints;   /* targetEdges.obj; 0 edges = use initEdges */
    int32_t nTargetEdges;      const int32_t* targetEdges;
    int32_t nSurfacePoints;    const double* surfacePoints;      /* [3 n] targetSurfaces.obj                 */
    int32_t nSurfaceTriangles; const int32_t* surfaceTriangles;  /* [3 n]                                    */
    const int32_t* isCornerPointIO;        /* [nPoints] or NULL                                              */
    const int32_t* isFeatureEdgePointIO;   /* [nPoints] or NULL                                              */
    double distanceTolerance;                  /* SM.C:1921: REL_TOL * min(mesh min edge length, layerEdgeLength) */
    double internalSmoothingBlendingFraction;  /* SM.C:1907, default 0                                       */
} smgpu_boundary_desc;
typedef struct smgpu_boundary_info {
    int32_t enabled;
    int32_t nCornerPoints, nFeatureEdgePoints, nSmoothingSurfacePoints, nFrozenSurfacePoints;   /* BPS.C:423-438 */
    int32_t nTargetEdgeStrings;                                                                /* SM.C:2171      */
} smgpu_boundary_info;
int smgpu_set_boundary_smoothing(smgpu_handle* h, const smgpu_boundary_desc* d, smgpu_boundary_info* info);
/* The same set-up in steps, for runs with a halo (-parallel; configure the halo and, if used, the layers first).  The host
 * performs the reference's reductions and syncTools::syncPointList calls between the steps (SM.C:1528-1538, 2218-2219):
 *   smgpu_boundary_stats on every rank; minimum of the edge lengths, bounding box over the ranks,
 *                                       perimeter = (max x - min x) + (max y - min y) + (max z + min z)   [SM.C:1538 as written]
 *   smgpu_boundary_begin(reduced values) on every rank
 *   2 x { step HOPS_SWEEP;  get F_HOPS, combine with max over the sharers (OBB.C:124-130), set F_HOPS }
 *   step TABLES
 *   step NORMALS_ACCUMULATE;  get F_NORMALS_COUNT, sum over the sharers in ascending rank order (OBB.C:184-198), set;
 *   step NORMALS_FINISH                      (both are no-ops when the layer set-up has produced the normals already)
 * Every later iteration the boundary point smoothing fields travel in the sendL / recvL records (SMGPU_HALO_L_DOUBLES). */
enum { SMGPU_BOUNDARY_HOPS_SWEEP = 0, SMGPU_BOUNDARY_TABLES = 1, SMGPU_BOUNDARY_NORMALS_ACCUMULATE = 2, SMGPU_BOUNDARY_NORMALS_FINISH = 3 };
enum { SMGPU_BOUNDARY_F_HOPS = 0,            /* 1 double per shared point (the hop count or -1)         */
       SMGPU_BOUNDARY_F_NORMALS_COUNT = 1 }; /* 4 doubles: local normal sum, local number of boundary faces */
int smgpu_boundary_stats(smgpu_handle* h, double* minEdgeLength, double* boundingBox /* [6]: min x, max x, min y, ... */);
int smgpu_boundary_begin(smgpu_handle* h, const smgpu_boundary_desc* d, double minEdgeLengthGlobal, double perimeterGlobal,
                         smgpu_boundary_info* info);
int smgpu_boundary_step(smgpu_handle* h, int32_t step);
int smgpu_boundary_shared(smgpu_handle* h, int32_t field, int32_t set, double* values);
/* the classification to persist as <time>/isCornerPoint and <time>/isFeatureEdgePoint (labelIOLists, SM.C:2039-2064) */
int smgpu_get_boundary_classification(smgpu_handle* h, int32_t* isCornerPoint, int32_t* isFeatureEdgePoint);
/* parity access, host only: the string index of every edge of an edge mesh (findEdgeMeshStrings BPS.C:557-587) */
int smgpu_debug_edge_strings(int32_t nPoints, int32_t nEdges, const int32_t* edges, int32_t* strings, int32_t* nStrings);
/* parity access: nearest intersections of n segments (6 doubles each: start, end) with the target surface */
int smgpu_debug_find_line(smgpu_handle* h, int32_t n, const double* segments, double* hitPoints, int32_t* hit);

/* nEngines engines compute on this handle's device at the same time (ranks sharing a GPU in a debugging run, the sub-domains of
 * one process): the persistent launch of the face-angle walk replay, whose workgroups all have to be resident at once, takes
 * 1/nEngines of its default size.  No reference counterpart (the reference runs one MPI rank per core).  Results do not
 * depend on it. */
int smgpu_set_device_share(smgpu_handle* h, int32_t nEngines);

/* The replay form of the face-angle freeze walk (SM.C:1347-1434) in use: -1 not decided yet, 0 one wave over the flag array
 * (few points outside the good angle range), 1 host replay (SMGPU_WALK=host), 2 compaction + causal fixed point (many); and how
 * often the automatic choice has changed since smgpu_set_params -- it follows the number of points outside the good range as
 * the run goes (results never depend on it); *lastCount = that number for the last iteration the GPU has closed (-1: none). */
int smgpu_debug_walk_mode(smgpu_handle* h, int32_t* mode, int32_t* switches, int32_t* lastCount);
/* how the LAST smgpu_iter_begin / mid / end went out (multi-rank): multiRole = geometry + pack and combine + smoothing + packF as
 * the two multi-role launches on tiles of the shared points (constraints off; 0 = one kernel per step), flagged = the host's
 * exchanges on the exchange stream ordered by flag words next to those launches, fixInside = k_shared_fix's work as a role of
 * the smoothing launch (peer-store transport) */
/* FNV-1a checksums of every array of the addressing in a fixed order (SMGPU_TOPO_CHECKSUMS words; [0] = sizes and maxima): the
 * engine's (built on the device where the mesh allows, csrc/topology_dev.hip) against smgpu_topology_checksums of the host build */
#define SMGPU_TOPO_CHECKSUMS 64
int smgpu_debug_addressing_checksums(smgpu_handle* h, uint64_t* out /* [SMGPU_TOPO_CHECKSUMS] */);
/* the same for the geometry tile tables as the kernels read them (device build, csrc/tiles_dev.hip, against SMGPU_DEVICE_TILES=0) */
int smgpu_debug_tile_checksums(smgpu_handle* h, uint64_t* out /* [SMGPU_TOPO_CHECKSUMS] */);
int smgpu_debug_halo_mode(smgpu_handle* h, int32_t* multiRole, int32_t* flagged, int32_t* fixInside);

/* self-test of the geometry kernel's range-tested square root / division fast paths (csrc/fpexact.hpp) against the plain
 * IEEE operators on n generated arguments (random, zeros, denormals, inf / nan, both ends of the exponent range) on the
 * given device; *mismatches = number of results whose bits differ (nan == nan).  0 is the only acceptable count. */
int smgpu_debug_selftest_fpexact(int32_t device, uint64_t seed, int64_t n, int64_t* mismatches);

/* ---- debug / parity access (device -> host copy of an internal field) -----------------------
 * name: "cellCentres" [3C], "faceCentres" [3F], "faceAreas" [3F], "newPoints" [3P] (proposal of the
 * last iteration before restore), "isFrozenPoint" [P], "edgeMinAngle"/"edgeMaxAngle" [E],
 * "pointMinAngle"/"pointMaxAngle" [P].  Values are converted to double.  Returns the element
 * count through *n; out may be NULL to query the size. */
int smgpu_debug_get_field(smgpu_handle* h, const char* name, double* out, int64_t* n);
/* kind: pointCells, pointPoints, pointFaces, pointEdges, edgeFaces, edgeCells, edges (2/row).
 * offsets/values may be NULL to query nnz. */
int smgpu_debug_get_addressing(smgpu_handle* h, const char* kind, int32_t* offsets, int32_t* values, int64_t* nnz);
/* Run one iteration up to (not including) restore/movePoints so "newPoints"/"isFrozenPoint"
 * can be compared with the oracle's phaseA/phaseB. */
int smgpu_debug_propose(smgpu_handle* h);


/* ---- host-only addressing build (no device needed; used by the CPU test-suite and hosts that
 * want the derived lists, e.g. to compute getMeshStats defaults before choosing a device) ------- */
typedef struct smgpu_topology smgpu_topology;
int smgpu_topology_create(const smgpu_mesh_desc* desc, smgpu_topology** out);
int smgpu_topology_get(smgpu_topology* t, const char* kind, int32_t* offsets, int32_t* values, int64_t* nnz);
int smgpu_topology_num_edges(smgpu_topology* t, int32_t* nEdges);
int smgpu_topology_checksums(smgpu_topology* t, uint64_t* out /* [SMGPU_TOPO_CHECKSUMS], see smgpu_debug_addressing_checksums */);
int smgpu_topology_destroy(smgpu_topology* t);

#ifdef __cplusplus
}
#endif
#endif /* SMGPU_H */
