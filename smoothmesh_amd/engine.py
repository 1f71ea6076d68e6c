"""Host-side mirror of the C-ABI (include/smgpu.h): one SmoothEngine = one rank's fvMesh in the
reference's loop (src/smoothMesh.C:2257-2437).  Parameter names are the reference's command-line
option names (SM.C:1642-1784), defaults as SM.C:1857-1918."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _ffi
from .mesh import PolyMesh


class SmgpuError(RuntimeError):
    """A non-zero status from the library (the reference would FatalError/abort here)."""


@dataclass
class SmoothParams:
    maxStepLength: float
    minEdgeLength: float
    relStepFrac: float = 0.5
    totalMinFreeze: bool = False
    edgeAngleConstraint: bool = True
    faceAngleConstraint: bool = True
    minAngle: float = 35.0
    maxAngle: float = 160.0


@dataclass
class LayerParams:
    """Boundary layer treatment options (SM.C:1749-1775), defaults SM.C:1892-1905."""
    layerPatches: tuple = ()                  # patch names; a name in double quotes is a regular expression
    layerMaxBlendingFraction: float = 0.3
    layerEdgeLength: Optional[float] = None   # None = minEdgeLength
    layerExpansionRatio: float = 1.3
    minLayers: int = 1
    maxLayers: int = 4


@dataclass
class BoundaryParams:
    """Boundary point smoothing inputs: the contents of constant/geometry/{initEdges,targetEdges,targetSurfaces}.obj
    (SM.C:1924-1926) and the options SM.C:1758-1770, 1907."""
    initEdges: tuple = None                   # (points (n,3), edges (m,2))
    targetSurfaces: tuple = None              # (points (n,3), triangles (m,3))
    targetEdges: tuple = None                 # None = the initial edges are the target (SM.C:2154-2160)
    smoothingPatches: tuple = ('".*"',)       # default: every patch (SM.C:1837-1840)
    internalSmoothingBlendingFraction: float = 0.0
    isCornerPointIO: object = None            # classification lists of a previous run (SM.C:2039-2077)
    isFeatureEdgePointIO: object = None


def patch_arrays(mesh: PolyMesh, layerPatches):
    """(start, size, kind, isLayer) of mesh.patches; kind 0 ordinary / 1 processor / 2 empty.  Selection as
    polyBoundaryMesh::patchSet (SM.C:1442-1471): a plain word matches a patch name, a quoted string is a regex."""
    import re
    pats = []
    for w in layerPatches:
        w = str(w)
        pats.append(re.compile(w[1:-1]) if len(w) >= 2 and w[0] == '"' and w[-1] == '"' else w)
    kinds = {"processor": 1, "empty": 2}
    start = np.array([p.startFace for p in mesh.patches], np.int32)
    size = np.array([p.nFaces for p in mesh.patches], np.int32)
    kind = np.array([kinds.get(p.type, 0) for p in mesh.patches], np.uint8)
    sel = np.array([any((q.fullmatch(p.name) is not None) if hasattr(q, "fullmatch") else (q == p.name) for q in pats)
                    for p in mesh.patches], np.uint8)
    return start, size, kind, sel


def default_params(meshMinEdgeLength: float, **over) -> SmoothParams:
    """SM.C:1861-1865: minEdgeLength = 0.5 * mesh min edge, maxStepLength = 0.3 * minEdgeLength."""
    minEdge = over.pop("minEdgeLength", 0.5 * meshMinEdgeLength)
    maxStep = over.pop("maxStepLength", 0.3 * minEdge)
    return SmoothParams(maxStepLength=maxStep, minEdgeLength=minEdge, **over)


def _p(a, t):
    return a.ctypes.data_as(t)


def make_desc(mesh: PolyMesh, isInternalPoint, isSmoothingSurfacePoint, device=0, stream=None):
    keep = dict(
        pts=np.ascontiguousarray(mesh.points, dtype=np.float64),
        fo=mesh.faceOffsets, fp=mesh.facePoints, ow=mesh.owner, ne=mesh.neighbour,
        ip=np.ascontiguousarray(isInternalPoint, dtype=np.uint8),
        sp=None if isSmoothingSurfacePoint is None else np.ascontiguousarray(isSmoothingSurfacePoint, dtype=np.uint8),
    )
    d = _ffi.MeshDesc()
    d.nPoints, d.nCells, d.nFaces, d.nInternalFaces = mesh.nPoints, mesh.nCells, mesh.nFaces, mesh.nInternalFaces
    d.points = _p(keep["pts"], _ffi.c_f64p)
    d.faceOffsets = _p(keep["fo"], _ffi.c_i32p)
    d.facePoints = _p(keep["fp"], _ffi.c_i32p)
    d.owner = _p(keep["ow"], _ffi.c_i32p)
    d.neighbour = _p(keep["ne"], _ffi.c_i32p)
    d.isInternalPoint = _p(keep["ip"], _ffi.c_u8p)
    d.isSmoothingSurfacePoint = _p(keep["sp"], _ffi.c_u8p) if keep["sp"] is not None else None
    d.device = device
    d.stream = stream if stream else None      # stream: None = library-owned stream; an int handle
    d.useCallerStream = 0 if stream is None else 1   # (0 = the HIP null stream) = run on the caller's
    return d, keep


TOPO_ARRAYS = ["sizes and maxima", "facePoints.off", "facePoints.val", "owner", "neighbour", "cellFacesGeom.off", "cellFacesGeom.val",
               "pointFaces.off", "pointFaces.val", "pfPrev", "pfNext", "pfPrevSlot", "pfNextSlot", "pointCells.off", "pointCells.val",
               "edges", "pointEdges.off", "pointEdges.val", "pointPoints", "edgeFaces.off", "edgeFaces.val", "edgeCells.off", "edgeCells.val",
               "ecFace0", "ecFace1", "ringFace", "ringCell", "edgeRingOk"]


class HostTopology:
    """Host-only addressing build (no GPU): the library's derived lists, for checks and hosts."""

    def __init__(self, mesh: PolyMesh):
        self._lib = _ffi.lib()
        d, self._keep = make_desc(mesh, np.ones(mesh.nPoints, np.uint8), None)
        self._h = C.c_void_p()
        if self._lib.smgpu_topology_create(C.byref(d), C.byref(self._h)):
            raise SmgpuError(self._lib.smgpu_last_error().decode())

    def addressing(self, kind):
        nnz = C.c_int64()
        k = kind.encode()
        if self._lib.smgpu_topology_get(self._h, k, None, None, C.byref(nnz)):
            raise SmgpuError(self._lib.smgpu_last_error().decode())
        vals = np.empty(nnz.value, np.int32)
        if kind == "edges":
            self._lib.smgpu_topology_get(self._h, k, None, _p(vals, _ffi.c_i32p), C.byref(nnz))
            return None, vals.reshape(-1, 2)
        rows = {"pointCells": "P", "pointPoints": "P", "pointEdges": "P", "pointFaces": "P", "pointFacePrev": "P",
                "pointFaceNext": "P", "edgeFaces": "E", "edgeCells": "E", "cellFacesGeom": "C"}[kind]
        n = {"P": self._keep["pts"].shape[0], "E": self.num_edges(), "C": int(self._keep["ow"].max()) + 1}[rows]
        off = np.empty(n + 1, np.int32)
        self._lib.smgpu_topology_get(self._h, k, _p(off, _ffi.c_i32p), _p(vals, _ffi.c_i32p), C.byref(nnz))
        return off, vals

    def num_edges(self):
        n = C.c_int32()
        self._lib.smgpu_topology_num_edges(self._h, C.byref(n))
        return n.value

    def checksums(self):
        """FNV-1a checksums of every array of the addressing, TOPO_ARRAYS order (include/smgpu.h smgpu_topology_checksums)"""
        out = (C.c_uint64 * 64)()
        if self._lib.smgpu_topology_checksums(self._h, out):
            raise SmgpuError(self._lib.smgpu_last_error().decode())
        return [int(x) for x in out][:len(TOPO_ARRAYS)]

    def close(self):
        if self._h:
            self._lib.smgpu_topology_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmoothEngine:
    def __init__(self, mesh: PolyMesh, isInternalPoint=None, isSmoothingSurfacePoint=None, device=0, stream=None):
        self._lib = _ffi.lib()
        self.mesh = mesh
        if isInternalPoint is None:
            isInternalPoint = mesh.find_internal_points()
        if isSmoothingSurfacePoint is None:
            isSmoothingSurfacePoint = mesh.smoothing_surface_points()
        self.isInternalPoint = np.ascontiguousarray(isInternalPoint, dtype=np.uint8)
        d, keep = make_desc(mesh, self.isInternalPoint, isSmoothingSurfacePoint, device, stream)
        self._h = C.c_void_p()
        self._check(self._lib.smgpu_create(C.byref(d), C.byref(self._h)))
        self.nPoints = mesh.nPoints
        self._keepalive = []

    def _check(self, rc):
        if rc:
            raise SmgpuError(self._lib.smgpu_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ---------------------------------------------------------------------------------
    def sizes(self):
        s = _ffi.Sizes()
        self._check(self._lib.smgpu_get_sizes(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in s._fields_}

    def mesh_stats(self):
        """getMeshStats, SM.C:1478-1541 -> (meshMinEdgeLength, meshMaxEdgeLength)."""
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.smgpu_mesh_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_params(self, p: SmoothParams):
        q = _ffi.Params(p.maxStepLength, p.relStepFrac, p.minEdgeLength, int(p.totalMinFreeze),
                        int(p.edgeAngleConstraint), int(p.faceAngleConstraint), p.minAngle, p.maxAngle)
        self._check(self._lib.smgpu_set_params(self._h, C.byref(q)))
        self.params = p

    def set_foam_variant(self, variant):
        """"com" (OpenFOAM.com v2312-v2506, default) or "org" (OpenFOAM.org 12): whose face / cell geometry formulas the
        cell centres follow (the reference builds against either, Allwmake:47)"""
        self._check(self._lib.smgpu_set_foam_variant(self._h, {"com": 0, "org": 1}[variant]))

    def set_sync_variant(self, variant):
        """"master" (default: globalMeshData::syncData -- the master's fold handed to every sharer) or "own" (every sharer folds the
        others' values onto its own): the syncTools::syncPointList model of the multi-rank magnitude folds (include/smgpu.h)"""
        self._check(self._lib.smgpu_set_sync_variant(self._h, {"master": 0, "own": 1}[variant]))

    # -- the loop ------------------------------------------------------------------------------
    def _layer_desc(self, lp: LayerParams, minEdgeLength: float):
        start, size, kind, sel = patch_arrays(self.mesh, lp.layerPatches)
        d = _ffi.LayerDesc()
        d.nPatches = len(start)
        d.patchStart, d.patchSize = _p(start, _ffi.c_i32p), _p(size, _ffi.c_i32p)
        d.patchKind, d.isLayerPatch = _p(kind, _ffi.c_u8p), _p(sel, _ffi.c_u8p)
        d.layerMaxBlendingFraction = lp.layerMaxBlendingFraction
        d.layerEdgeLength = minEdgeLength if lp.layerEdgeLength is None else lp.layerEdgeLength
        d.layerExpansionRatio = lp.layerExpansionRatio
        d.minLayers, d.maxLayers = lp.minLayers, lp.maxLayers
        return d, (start, size, kind, sel)

    # step-wise set-up for runs with a halo (see include/smgpu.h, smgpu_layers_begin)
    LAYERS_HOPS_SWEEP, LAYERS_NORMALS_ACCUMULATE, LAYERS_NORMALS_FINISH, LAYERS_PROPAGATE_SWEEP, LAYERS_FINISH = range(5)
    LAYERS_F_HOPS, LAYERS_F_NORMALS_COUNT, LAYERS_F_NORMALS = range(3)
    _LAYER_FIELD_WIDTH = (1, 4, 3)

    def layers_begin(self, lp: LayerParams, minEdgeLength: float):
        d, keep = self._layer_desc(lp, minEdgeLength)
        on, it = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.smgpu_layers_begin(self._h, C.byref(d), C.byref(on), C.byref(it)))
        return bool(on.value), it.value

    def layers_step(self, step, arg=0):
        self._check(self._lib.smgpu_layers_step(self._h, int(step), int(arg)))

    def layers_shared_get(self, field):
        v = np.zeros((self._nShared, self._LAYER_FIELD_WIDTH[field]), np.float64)
        if self._nShared:
            self._check(self._lib.smgpu_layers_shared(self._h, int(field), 0, _p(v, _ffi.c_f64p)))
        return v

    def layers_shared_set(self, field, values):
        v = np.ascontiguousarray(values, np.float64).reshape(self._nShared, self._LAYER_FIELD_WIDTH[field])
        if self._nShared:
            self._check(self._lib.smgpu_layers_shared(self._h, int(field), 1, _p(v, _ffi.c_f64p)))

    def l_doubles(self):
        """doubles per slot of the exchange-L records in use (6 with the layer treatment only, 14 with boundary smoothing)"""
        n = C.c_int32(0)
        self._check(self._lib.smgpu_halo_l_doubles(self._h, C.byref(n)))
        return n.value

    def set_layers(self, lp: LayerParams, minEdgeLength: float):
        """Enable the boundary layer treatment on lp.layerPatches (serial runs); returns the reference's
        doLayerTreatment.  Call after construction, before iterating."""
        d, keep = self._layer_desc(lp, minEdgeLength)
        on = C.c_int32(0)
        self._check(self._lib.smgpu_set_layers(self._h, C.byref(d), C.byref(on)))
        return bool(on.value)

    def _boundary_desc(self, bp: "BoundaryParams", minEdgeLength: float, layerEdgeLength=None, meshMinEdgeLength=None):
        REL_TOL = 1e-4                                                     # COM.H:20
        lel = minEdgeLength if layerEdgeLength is None else layerEdgeLength
        mn = self.mesh_stats()[0] if meshMinEdgeLength is None else meshMinEdgeLength
        tol = REL_TOL * min(mn, lel)                                       # SM.C:1921
        start, size, kind, sel = patch_arrays(self.mesh, bp.smoothingPatches)
        def pe(m, w):
            if m is None:
                return np.zeros((0, 3), np.float64), np.zeros((0, w), np.int32)
            return np.ascontiguousarray(m[0], np.float64).reshape(-1, 3), np.ascontiguousarray(m[1], np.int32).reshape(-1, w)
        ip, ie = pe(bp.initEdges, 2); tp, te = pe(bp.targetEdges, 2); sp, st = pe(bp.targetSurfaces, 3)
        cio = None if bp.isCornerPointIO is None else np.ascontiguousarray(bp.isCornerPointIO, np.int32)
        fio = None if bp.isFeatureEdgePointIO is None else np.ascontiguousarray(bp.isFeatureEdgePointIO, np.int32)
        d = _ffi.BoundaryDesc()
        d.nPatches = len(start)
        d.patchStart, d.patchSize = _p(start, _ffi.c_i32p), _p(size, _ffi.c_i32p)
        d.patchKind, d.isSmoothingPatch = _p(kind, _ffi.c_u8p), _p(sel, _ffi.c_u8p)
        d.nInitEdgePoints, d.initEdgePoints, d.nInitEdges, d.initEdges = len(ip), _p(ip, _ffi.c_f64p), len(ie), _p(ie, _ffi.c_i32p)
        d.nTargetEdgePoints, d.targetEdgePoints, d.nTargetEdges, d.targetEdges = len(tp), _p(tp, _ffi.c_f64p), len(te), _p(te, _ffi.c_i32p)
        d.nSurfacePoints, d.surfacePoints, d.nSurfaceTriangles, d.surfaceTriangles = len(sp), _p(sp, _ffi.c_f64p), len(st), _p(st, _ffi.c_i32p)
        d.isCornerPointIO = None if cio is None else _p(cio, _ffi.c_i32p)
        d.isFeatureEdgePointIO = None if fio is None else _p(fio, _ffi.c_i32p)
        d.distanceTolerance = tol
        d.internalSmoothingBlendingFraction = bp.internalSmoothingBlendingFraction
        return d, (start, size, kind, sel, ip, ie, tp, te, sp, st, cio, fio)

    def set_boundary_smoothing(self, bp: "BoundaryParams", minEdgeLength: float, layerEdgeLength=None):
        """Enable the boundary point smoothing (serial runs; after set_layers when both are used).  minEdgeLength is
        the -minEdgeLength option value (the default of layerEdgeLength, SM.C:1895).  Returns a dict with the
        reference's doBoundarySmoothing ("enabled") and the classification summary (BPS.C:423-438)."""
        d, keep = self._boundary_desc(bp, minEdgeLength, layerEdgeLength)
        info = _ffi.BoundaryInfo()
        self._check(self._lib.smgpu_set_boundary_smoothing(self._h, C.byref(d), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ffi.BoundaryInfo._fields_}

    # step-wise set-up for runs with a halo (see include/smgpu.h, smgpu_boundary_begin)
    BOUNDARY_HOPS_SWEEP, BOUNDARY_TABLES, BOUNDARY_NORMALS_ACCUMULATE, BOUNDARY_NORMALS_FINISH = range(4)
    BOUNDARY_F_HOPS, BOUNDARY_F_NORMALS_COUNT = range(2)
    _BOUNDARY_FIELD_WIDTH = (1, 4)

    def boundary_stats(self):
        """(minimum edge length, bounding box [min x, max x, min y, max y, min z, max z]) of this rank (SM.C:1478-1526)"""
        mn, bb = C.c_double(0), np.zeros(6, np.float64)
        self._check(self._lib.smgpu_boundary_stats(self._h, C.byref(mn), _p(bb, _ffi.c_f64p)))
        return mn.value, bb

    def boundary_begin(self, bp, minEdgeLength, minEdgeGlobal, perimeterGlobal, layerEdgeLength=None):
        d, keep = self._boundary_desc(bp, minEdgeLength, layerEdgeLength, meshMinEdgeLength=minEdgeGlobal)
        info = _ffi.BoundaryInfo()
        self._check(self._lib.smgpu_boundary_begin(self._h, C.byref(d), float(minEdgeGlobal), float(perimeterGlobal), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ffi.BoundaryInfo._fields_}

    def boundary_step(self, step):
        self._check(self._lib.smgpu_boundary_step(self._h, int(step)))

    def boundary_shared_get(self, field):
        v = np.zeros((self._nShared, self._BOUNDARY_FIELD_WIDTH[field]), np.float64)
        if self._nShared:
            self._check(self._lib.smgpu_boundary_shared(self._h, int(field), 0, _p(v, _ffi.c_f64p)))
        return v

    def boundary_shared_set(self, field, values):
        v = np.ascontiguousarray(values, np.float64).reshape(self._nShared, self._BOUNDARY_FIELD_WIDTH[field])
        if self._nShared:
            self._check(self._lib.smgpu_boundary_shared(self._h, int(field), 1, _p(v, _ffi.c_f64p)))

    def boundary_classification(self):
        """(isCornerPoint, isFeatureEdgePoint) as the labelIOLists the reference writes (SM.C:2039-2064)."""
        a, b = np.zeros(self.nPoints, np.int32), np.zeros(self.nPoints, np.int32)
        self._check(self._lib.smgpu_get_boundary_classification(self._h, _p(a, _ffi.c_i32p), _p(b, _ffi.c_i32p)))
        return a, b

    def debug_find_line(self, starts, ends):
        """nearest intersections of the segments with the target surface: (hit mask, hit points)"""
        seg = np.ascontiguousarray(np.concatenate([np.asarray(starts, np.float64).reshape(-1, 3),
                                                   np.asarray(ends, np.float64).reshape(-1, 3)], axis=1))
        n = len(seg)
        out, hit = np.zeros((n, 3), np.float64), np.zeros(n, np.int32)
        self._check(self._lib.smgpu_debug_find_line(self._h, n, _p(seg, _ffi.c_f64p), _p(out, _ffi.c_f64p), _p(hit, _ffi.c_i32p)))
        return hit.astype(bool), out

    def iterate(self, centroidalIters: int, relTol: float = 0.02):
        """Returns (nDone, residuals[nDone], nFrozenPoints[nDone]) -- the values of the reference's
        per-iteration log line (SM.C:2396)."""
        stats = (_ffi.IterStats * max(centroidalIters, 1))()
        nDone = C.c_int32()
        self._check(self._lib.smgpu_iterate(self._h, centroidalIters, relTol, stats, C.byref(nDone)))
        n = nDone.value
        res = np.array([stats[i].residual for i in range(n)], dtype=np.float64)
        frz = np.array([stats[i].nFrozenPoints for i in range(n)], dtype=np.int64)
        self.last_near_ties = np.array([stats[i].nNearTies for i in range(n)], dtype=np.int64)     # per iteration of this call
        return n, res, frz

    def near_ties(self):
        """Near-tie census since the engine was created (include/smgpu.h, smgpu_get_near_ties): {"total", "edge_angle" (SM.C:923),
        "good_range" (SM.C:1367), "walk" (SM.C:1391-1394 / 1421-1424)} -- angle comparisons whose two sides were 1 .. 4 ulp apart,
        i.e. decisions the reference's acos could have taken the other way.  All zero in a normal run."""
        out = (C.c_int64 * 4)()
        self._check(self._lib.smgpu_get_near_ties(self._h, out))
        return {"total": int(out[0]), "edge_angle": int(out[1]), "good_range": int(out[2]), "walk": int(out[3])}

    def check_error(self):
        """wait for the engine's stream and raise SmgpuError for any error word a kernel has raised (include/smgpu.h)"""
        self._check(self._lib.smgpu_check_error(self._h))

    def get_points(self):
        out = np.empty((self.nPoints, 3), np.float64)
        self._check(self._lib.smgpu_get_points(self._h, _p(out, _ffi.c_f64p)))
        return out

    def set_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        assert pts.shape == (self.nPoints, 3)
        self._check(self._lib.smgpu_set_points(self._h, _p(pts, _ffi.c_f64p)))

    # -- timing --------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._check(self._lib.smgpu_enable_timing(self._h, int(on)))

    def reset_counters(self):
        self._check(self._lib.smgpu_reset_counters(self._h))

    def counters(self):
        c = _ffi.Counters()
        self._check(self._lib.smgpu_get_counters(self._h, C.byref(c)))
        return [dict(name=c.name[i].decode(), ms=c.ms[i], launches=c.launches[i], algoBytesPerLaunch=c.algoBytesPerLaunch[i],
                     algoF64OpsPerLaunch=c.algoF64OpsPerLaunch[i])
                for i in range(c.nKernels)]

    # -- multi-rank ----------------------------------------------------------------------------
    def halo_configure(self, sharedLocal, sendShared, nRecv, combOffsets, combSlots, sendA, recvA, sendF, recvF, localStats,
                       exchangeStream=None, sendL=None, recvL=None):
        """Pointers are raw device addresses (ints); exchangeStream: raw hipStream_t (int, 0 = null stream) the
        caller enqueues its exchanges on, or None = the engine's own stream; see smgpu_halo_desc."""
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (sharedLocal, sendShared, combOffsets, combSlots)]
        d = _ffi.HaloDesc()
        d.nShared = len(keep[0]); d.sharedLocal = _p(keep[0], _ffi.c_i32p)
        d.nSend = len(keep[1]); d.sendShared = _p(keep[1], _ffi.c_i32p)
        d.nRecv = int(nRecv); d.combOffsets = _p(keep[2], _ffi.c_i32p); d.combSlots = _p(keep[3], _ffi.c_i32p)
        d.sendA, d.recvA, d.sendF, d.recvF, d.localStats = sendA, recvA, sendF, recvF, localStats
        d.sendL, d.recvL = sendL, recvL
        self._nShared = len(keep[0])
        d.useExchangeStream = 0 if exchangeStream is None else 1
        d.exchangeStream = exchangeStream or None
        self._check(self._lib.smgpu_halo_configure(self._h, C.byref(d)))

    def set_push(self, peerCount, remoteBase, myIndexAtPeer, peerRecvA, peerRecvL, peerRecvF, peerFlags, localFlags):
        """peer-store transport (smgpu_halo_set_push): arrays per peer in ascending rank order; pointers = raw device addresses"""
        n = len(peerCount)
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (peerCount, remoteBase, myIndexAtPeer)]
        arr = lambda v: (C.c_void_p * max(n, 1))(*[C.c_void_p(int(x) if x else None) for x in v])
        ptrs = [arr(v) for v in (peerRecvA, peerRecvL, peerRecvF, peerFlags)]
        d = _ffi.PushDesc()
        d.nPeers = n
        d.peerCount, d.remoteBase, d.myIndexAtPeer = (_p(k, _ffi.c_i32p) for k in keep)
        d.peerRecvA, d.peerRecvL, d.peerRecvF, d.peerFlags = (C.cast(a, C.POINTER(C.c_void_p)) for a in ptrs)
        d.localFlags = localFlags
        self._check(self._lib.smgpu_halo_set_push(self._h, C.byref(d)))

    def clear_push(self):
        self._check(self._lib.smgpu_halo_set_push(self._h, None))

    def set_exchange_stream(self, exchangeStream):
        """None = exchanges are enqueued on the engine's stream; int = raw hipStream_t they are enqueued on"""
        self._check(self._lib.smgpu_halo_set_exchange_stream(self._h, 0 if exchangeStream is None else 1, exchangeStream or None))

    def stream(self):
        """raw hipStream_t handle (int) of the engine's stream"""
        out = C.c_void_p()
        self._check(self._lib.smgpu_get_stream(self._h, C.byref(out)))
        return out.value or 0

    def set_stats_history(self, ptr, capacity):
        """device array of 2*capacity doubles that iter_end fills record by record (None switches it off)"""
        self._check(self._lib.smgpu_halo_set_stats_history(self._h, ptr, int(capacity)))

    def iter_begin(self):
        self._check(self._lib.smgpu_iter_begin(self._h))

    def iter_interior(self):
        self._check(self._lib.smgpu_iter_interior(self._h))

    def iter_mid(self):
        self._check(self._lib.smgpu_iter_mid(self._h))

    def iter_ahead(self):
        self._check(self._lib.smgpu_iter_ahead(self._h))

    def iter_end(self):
        self._check(self._lib.smgpu_iter_end(self._h))

    # -- debug / parity ------------------------------------------------------------------------
    def debug_walk_mode(self):
        """(replay form of the face-angle walk in use, number of automatic changes since set_params)"""
        mode, sw, cnt = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.smgpu_debug_walk_mode(self._h, C.byref(mode), C.byref(sw), C.byref(cnt)))
        self.last_active_count = cnt.value
        return mode.value, sw.value

    def debug_addressing_checksums(self):
        """checksums of the engine's addressing (built on the device where the mesh allows), TOPO_ARRAYS order"""
        out = (C.c_uint64 * 64)()
        self._check(self._lib.smgpu_debug_addressing_checksums(self._h, out))
        return [int(x) for x in out][:len(TOPO_ARRAYS)]

    def debug_tile_checksums(self):
        """checksums of the geometry tile tables as the kernels read them (include/smgpu.h smgpu_debug_tile_checksums)"""
        out = (C.c_uint64 * 64)()
        self._check(self._lib.smgpu_debug_tile_checksums(self._h, out))
        return [int(x) for x in out]

    def debug_halo_mode(self):
        """how the last multi-rank iteration went out: {"multi_role", "flagged", "fix_inside"} (include/smgpu.h)"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.smgpu_debug_halo_mode(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"multi_role": bool(a.value), "flagged": bool(b.value), "fix_inside": bool(c.value)}

    def set_device_share(self, n_engines):
        """n_engines engines compute on this device at the same time: the persistent walk replay takes its share of the chip"""
        self._check(self._lib.smgpu_set_device_share(self._h, int(n_engines)))

    def debug_propose(self):
        self._check(self._lib.smgpu_debug_propose(self._h))

    def debug_field(self, name):
        n = C.c_int64()
        self._check(self._lib.smgpu_debug_get_field(self._h, name.encode(), None, C.byref(n)))
        out = np.empty(n.value, np.float64)
        self._check(self._lib.smgpu_debug_get_field(self._h, name.encode(), _p(out, _ffi.c_f64p), C.byref(n)))
        return out
