"""ctypes wrapper over oracle/liboracle.so.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by smoothmesh_amd/.
PARITY UNPINNED (see oracle/smooth_oracle.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOOTHMESH_ORACLE_LIB", os.path.join(_HERE, "liboracle.so"))   # override: sanitizer builds
_lib = None
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


SYNC_VARIANTS = {"master": 0, "own": 1}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_create.restype = C.c_void_p
        l.orc_create.argtypes = [C.c_int] * 4 + [f64p, i32p, i32p, i32p, i32p, u8p, u8p]
        l.orc_destroy.argtypes = [C.c_void_p]
        l.orc_set_acos_variant.argtypes = [C.c_int]
        l.orc_get_acos_variant.restype = C.c_int
        l.orc_acos_census_enable.argtypes = [C.c_int]
        l.orc_acos_census.argtypes = [C.POINTER(C.c_longlong)]
        l.orc_acos_census_window.argtypes = [C.c_longlong]
        l.orc_acos_census_near.argtypes = [C.POINTER(C.c_longlong)]
        l.orc_acos.restype = C.c_double
        l.orc_acos.argtypes = [C.c_double, C.c_int]
        l.orc_set_foam_variant.argtypes = [C.c_void_p, C.c_int]
        l.orc_set_sync_variant.argtypes = [C.c_void_p, C.c_int]
        l.orc_multi_set_sync_variant.argtypes = [C.c_void_p, C.c_int]
        l.orc_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        l.orc_mesh_stats.argtypes = [C.c_void_p, f64p, f64p]
        l.orc_iterate.restype = C.c_int
        l.orc_iterate.argtypes = [C.c_void_p, C.c_int, C.c_double, f64p, i32p]
        l.orc_last_error.restype = C.c_char_p
        l.orc_last_error.argtypes = [C.c_void_p]
        l.orc_set_points.argtypes = [C.c_void_p, f64p]
        l.orc_num_edges.restype = C.c_int
        l.orc_num_edges.argtypes = [C.c_void_p]
        for n in ("orc_phaseA", "orc_phaseB", "orc_phaseC", "orc_commit"):
            getattr(l, n).argtypes = [C.c_void_p]
        l.orc_get_field.restype = C.c_longlong
        l.orc_get_field.argtypes = [C.c_void_p, C.c_char_p, f64p]
        l.orc_get_addressing.restype = C.c_longlong
        l.orc_get_addressing.argtypes = [C.c_void_p, C.c_char_p, i32p, i32p]
        l.orc_edgeEdgeAngle.restype = C.c_double
        l.orc_edgeEdgeAngle.argtypes = [f64p, f64p, f64p]
        l.orc_calcEdgeCenterEdgeAngle.restype = C.c_double
        l.orc_calcEdgeCenterEdgeAngle.argtypes = [f64p, f64p, f64p]
        l.orc_isCloserPoint.restype = C.c_int
        l.orc_isCloserPoint.argtypes = [f64p, f64p]
        l.orc_multi_create.restype = C.c_void_p
        l.orc_multi_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.orc_multi_destroy.argtypes = [C.c_void_p]
        l.orc_multi_set_shared.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p]
        l.orc_multi_iterate.restype = C.c_int
        l.orc_multi_iterate.argtypes = [C.c_void_p, C.c_int, C.c_double, f64p, i32p]
        l.orc_halo_packA.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int, i32p, f64p]
        l.orc_halo_combineA.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, f64p]
        l.orc_halo_packF.argtypes = [C.c_void_p, i32p, C.c_int, i32p, i32p]
        l.orc_halo_orF.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p]
        l.orc_local_stats.argtypes = [C.c_void_p, f64p]
        l.orc_halo_packL.argtypes = [C.c_void_p, i32p, C.c_int, i32p, f64p]
        l.orc_halo_combineL.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, f64p]
        l.orc_layers_begin.restype = C.c_int
        l.orc_layers_begin.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, u8p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        l.orc_layers_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.orc_layers_shared.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, f64p]
        l.orc_setup_layers.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, u8p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        l.orc_multi_setup_layers.argtypes = [C.c_void_p, i32p, i32p, i32p, i32p, u8p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        l.orc_layers_enabled.restype = C.c_int
        l.orc_layers_enabled.argtypes = [C.c_void_p]
        l.orc_get_layer_fields.argtypes = [C.c_void_p, i32p, i32p, f64p, u8p, u8p]
        l.orc_setup_boundary.restype = C.c_int
        l.orc_setup_boundary.argtypes = ([C.c_void_p, C.c_int, i32p, i32p, i32p, u8p, u8p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
                                         + [C.c_int, f64p, C.c_int, i32p] * 2 + [C.c_int, f64p, C.c_int, i32p, i32p, i32p, C.c_double])
        l.orc_get_boundary_fields.argtypes = [C.c_void_p, u8p, u8p, u8p, u8p, f64p, i32p, i32p, i32p, f64p]
        l.orc_get_edge_strings.restype = C.c_int
        l.orc_get_edge_strings.argtypes = [C.c_void_p, i32p]
        l.orc_multi_setup_boundary.restype = C.c_int
        l.orc_multi_setup_boundary.argtypes = ([C.c_void_p, i32p, i32p, i32p, i32p, u8p, u8p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
                                               + [C.c_int, f64p, C.c_int, i32p] * 3 + [C.c_double])
        l.orc_edge_strings.restype = C.c_int
        l.orc_edge_strings.argtypes = [C.c_int, C.c_int, i32p, i32p]
        l.orc_find_line.restype = C.c_int
        l.orc_find_line.argtypes = [C.c_void_p, f64p, f64p, f64p]
        _lib = l
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _v(x):
    a = np.ascontiguousarray(x, dtype=np.float64)
    return a, _p(a, f64p)


class Oracle:
    """One Domain of the restatement (one rank's mesh)."""

    def __init__(self, mesh, isInternalPoint=None, isSmoothingSurfacePoint=None):
        self._lib = lib()
        if isInternalPoint is None:
            isInternalPoint = mesh.find_internal_points()
        ip = np.ascontiguousarray(isInternalPoint, dtype=np.uint8)
        sp = None if isSmoothingSurfacePoint is None else np.ascontiguousarray(isSmoothingSurfacePoint, dtype=np.uint8)
        pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
        self.nPoints, self.nCells, self.nFaces = mesh.nPoints, mesh.nCells, mesh.nFaces
        self._h = self._lib.orc_create(mesh.nPoints, mesh.nCells, mesh.nFaces, mesh.nInternalFaces, _p(pts, f64p),
                                       _p(mesh.faceOffsets, i32p), _p(mesh.facePoints, i32p), _p(mesh.owner, i32p),
                                       _p(mesh.neighbour, i32p), _p(ip, u8p), _p(sp, u8p) if sp is not None else None)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_foam_variant(self, variant):
        """"com" (default) / "org": OpenFOAM line whose primitiveMesh geometry formulas are used (smooth_oracle.hpp)"""
        self._lib.orc_set_foam_variant(self._h, {"com": 0, "org": 1}[variant])

    def set_sync_variant(self, variant):
        """"master" (default: globalMeshData::syncData, every sharer receives the master's fold) / "own" (every sharer folds the
        others' values onto its own): syncPointList model of the rank-engine combines (smooth_oracle.cpp)"""
        self._lib.orc_set_sync_variant(self._h, SYNC_VARIANTS[variant])

    def set_params(self, p):
        self._lib.orc_set_params(self._h, p.maxStepLength, p.relStepFrac, p.minEdgeLength, int(p.totalMinFreeze),
                                 int(p.edgeAngleConstraint), int(p.faceAngleConstraint), p.minAngle, p.maxAngle)

    def setup_layers(self, start, size, kind, isLayer, layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio,
                     minLayers, maxLayers):
        """boundary layer treatment set-up SM.C:2186-2221 (serial); returns doLayerTreatment"""
        st, sz = np.ascontiguousarray(start, np.int32), np.ascontiguousarray(size, np.int32)
        kd, il = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(isLayer, np.uint8)
        self._lib.orc_setup_layers(self._h, len(st), _p(st, i32p), _p(sz, i32p), _p(kd, i32p), _p(il, u8p),
                                   layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio, minLayers, maxLayers)
        return bool(self._lib.orc_layers_enabled(self._h))

    def setup_boundary(self, start, size, kind, isLayer, isSmoothing, layer, initEdges, targetEdges, surf, cornerIO=None,
                       featureIO=None, internalSmoothingBlendingFraction=0.0):
        """boundary point smoothing set-up SM.C:2080-2253 (serial).  layer = (layerMaxBlendingFraction, layerEdgeLength,
        layerExpansionRatio, minLayers, maxLayers); initEdges / targetEdges = (points (n,3), edges (m,2)) or None;
        surf = (points (n,3), triangles (m,3)).  Returns doBoundarySmoothing."""
        st, sz = np.ascontiguousarray(start, np.int32), np.ascontiguousarray(size, np.int32)
        kd, il = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(isLayer, np.uint8)
        ism = np.ascontiguousarray(isSmoothing, np.uint8)
        def pe(m):
            if m is None:
                return np.zeros((0, 3), np.float64), np.zeros((0, 2), np.int32)
            return np.ascontiguousarray(m[0], np.float64).reshape(-1, 3), np.ascontiguousarray(m[1], np.int32)
        ip, ie = pe(initEdges); tp, te = pe(targetEdges); sp, stri = pe(surf)
        cio = None if cornerIO is None else np.ascontiguousarray(cornerIO, np.int32)
        fio = None if featureIO is None else np.ascontiguousarray(featureIO, np.int32)
        rc = self._lib.orc_setup_boundary(self._h, len(st), _p(st, i32p), _p(sz, i32p), _p(kd, i32p), _p(il, u8p), _p(ism, u8p),
                                          float(layer[0]), float(layer[1]), float(layer[2]), int(layer[3]), int(layer[4]),
                                          len(ip), _p(ip, f64p), len(ie), _p(ie, i32p), len(tp), _p(tp, f64p), len(te), _p(te, i32p),
                                          len(sp), _p(sp, f64p), len(stri), _p(stri, i32p),
                                          None if cio is None else _p(cio, i32p), None if fio is None else _p(fio, i32p),
                                          float(internalSmoothingBlendingFraction))
        if rc < 0:
            raise RuntimeError(self._lib.orc_last_error(self._h).decode())
        return bool(rc)

    def boundary_fields(self):
        n = self.nPoints
        u8 = lambda: np.empty(n, np.uint8)
        i32 = lambda: np.empty(n, np.int32)
        co, fe, ss, sh = u8(), u8(), u8(), u8()
        cp, nrm = np.empty((n, 3), np.float64), np.empty((n, 3), np.float64)
        ps, im, hs = i32(), i32(), i32()
        self._lib.orc_get_boundary_fields(self._h, _p(co, u8p), _p(fe, u8p), _p(ss, u8p), _p(sh, u8p), _p(cp, f64p), _p(ps, i32p),
                                          _p(im, i32p), _p(hs, i32p), _p(nrm, f64p))
        ne = self._lib.orc_get_edge_strings(self._h, None)
        es = np.empty(max(ne, 1), np.int32)
        self._lib.orc_get_edge_strings(self._h, _p(es, i32p))
        return dict(isCornerPoint=co, isFeatureEdgePoint=fe, isSmoothingSurfacePoint=ss, isSharpEdgePoint=sh, cornerPoints=cp,
                    pointStrings=ps, innerMap=im, hopsToSmoothingBoundary=hs, normals=nrm, targetEdgeStrings=es[:ne])

    def find_line(self, start, end):
        a, pa = _v(start); b, pb = _v(end)
        out = np.empty(3, np.float64)
        hit = self._lib.orc_find_line(self._h, pa, pb, _p(out, f64p))
        return bool(hit), out

    def layer_fields(self):
        n = self.nPoints
        hops, omap = np.empty(n, np.int32), np.empty(n, np.int32)
        nrm = np.empty((n, 3), np.float64)
        con, lsp = np.empty(n, np.uint8), np.empty(n, np.uint8)
        self._lib.orc_get_layer_fields(self._h, _p(hops, i32p), _p(omap, i32p), _p(nrm, f64p), _p(con, u8p), _p(lsp, u8p))
        return dict(hops=hops, outerMap=omap, normals=nrm, isConnectedToInternalPoint=con, isLayerSurfacePoint=lsp)

    def mesh_stats(self):
        a, b = C.c_double(), C.c_double()
        self._lib.orc_mesh_stats(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def iterate(self, nIters, relTol=0.02):
        res = np.zeros(max(nIters, 1), np.float64)
        frz = np.zeros(max(nIters, 1), np.int32)
        n = self._lib.orc_iterate(self._h, nIters, relTol, _p(res, f64p), _p(frz, i32p))
        if n < 0:
            raise RuntimeError(self._lib.orc_last_error(self._h).decode())
        return n, res[:n].copy(), frz[:n].astype(np.int64)

    def set_points(self, pts):
        a, p = _v(pts)
        self._lib.orc_set_points(self._h, p)

    def phaseA(self): self._lib.orc_phaseA(self._h)
    def phaseB(self): self._lib.orc_phaseB(self._h)
    def phaseC(self): self._lib.orc_phaseC(self._h)
    def commit(self): self._lib.orc_commit(self._h)

    def error(self):
        return self._lib.orc_last_error(self._h).decode()

    def field(self, name):
        n = self._lib.orc_get_field(self._h, name.encode(), None)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.float64)
        self._lib.orc_get_field(self._h, name.encode(), _p(out, f64p))
        return out

    def points(self):
        return self.field("points").reshape(-1, 3)

    def num_edges(self):
        return self._lib.orc_num_edges(self._h)

    def addressing(self, kind):
        nnz = self._lib.orc_get_addressing(self._h, kind.encode(), None, None)
        if nnz < 0:
            raise KeyError(kind)
        vals = np.empty(nnz, np.int32)
        if kind == "edges":
            self._lib.orc_get_addressing(self._h, kind.encode(), None, _p(vals, i32p))
            return None, vals.reshape(-1, 2)
        rows = {"pointCells": self.nPoints, "pointFaces": self.nPoints, "pointEdges": self.nPoints,
                "pointPoints": self.nPoints, "edgeFaces": self.num_edges(), "edgeCells": self.num_edges(),
                "cellFaces": self.nCells}[kind]
        off = np.empty(rows + 1, np.int32)
        self._lib.orc_get_addressing(self._h, kind.encode(), _p(off, i32p), _p(vals, i32p))
        return off, vals


class OracleRankEngine:
    """An oracle Domain behind the engine interface smoothmesh_amd.halo drives (mesh_stats, set_params,
    halo_configure, iter_begin/mid/end, get_points) -- lets the CPU tests run the product's multi-rank
    host logic (slot tables, torch.distributed exchange, stop rule) under gloo with the oracle standing
    in for the HIP kernels.  Buffers arrive as raw host addresses (CPU torch tensors)."""

    def __init__(self, mesh):
        self.o = Oracle(mesh)
        self.o_mesh = mesh
        self._lib = lib()
        if os.environ.get("SMGPU_SYNC_VARIANT", "") == "own":      # the variable the product's engine reads
            self.o.set_sync_variant("own")

    def mesh_stats(self):
        return self.o.mesh_stats()

    def set_params(self, p):
        self.o.set_params(p)

    def set_sync_variant(self, variant):
        self.o.set_sync_variant(variant)

    def halo_configure(self, sharedLocal, sendShared, nRecv, combOffsets, combSlots, sendA, recvA, sendF, recvF, localStats,
                       exchangeStream=None, sendL=None, recvL=None):
        self.layers = False
        self.ptrL = (C.cast(sendL, f64p), C.cast(recvL, f64p)) if sendL else None
        self.sharedLocal = np.ascontiguousarray(sharedLocal, np.int32)
        self.sendShared = np.ascontiguousarray(sendShared, np.int32)
        self.combOffsets = np.ascontiguousarray(combOffsets, np.int32)
        self.combSlots = np.ascontiguousarray(combSlots, np.int32)
        self.ptr = dict(sendA=C.cast(sendA, f64p), recvA=C.cast(recvA, f64p), sendF=C.cast(sendF, i32p),
                        recvF=C.cast(recvF, i32p), localStats=C.cast(localStats, f64p))

    # step-wise boundary layer set-up, same interface as smoothmesh_amd.engine.SmoothEngine
    LAYERS_HOPS_SWEEP, LAYERS_NORMALS_ACCUMULATE, LAYERS_NORMALS_FINISH, LAYERS_PROPAGATE_SWEEP, LAYERS_FINISH = range(5)
    LAYERS_F_HOPS, LAYERS_F_NORMALS_COUNT, LAYERS_F_NORMALS = range(3)
    _LAYER_FIELD_WIDTH = (1, 4, 3)

    def layers_begin(self, lp, minEdgeLength):
        from smoothmesh_amd import patch_arrays
        st, sz, kd, il = patch_arrays(self.o_mesh, lp.layerPatches)
        st, sz = np.ascontiguousarray(st, np.int32), np.ascontiguousarray(sz, np.int32)
        kd, il = np.ascontiguousarray(kd, np.int32), np.ascontiguousarray(il, np.uint8)
        on = self._lib.orc_layers_begin(self.o._h, len(st), _p(st, i32p), _p(sz, i32p), _p(kd, i32p), _p(il, u8p),
                                        lp.layerMaxBlendingFraction, minEdgeLength if lp.layerEdgeLength is None else lp.layerEdgeLength,
                                        lp.layerExpansionRatio, lp.minLayers, lp.maxLayers)
        return bool(on), lp.maxLayers + 1

    def layers_step(self, step, arg=0):
        self._lib.orc_layers_step(self.o._h, int(step), int(arg))
        if step == self.LAYERS_FINISH:
            self.layers = True

    def layers_shared_get(self, field):
        v = np.zeros((len(self.sharedLocal), self._LAYER_FIELD_WIDTH[field]), np.float64)
        self._lib.orc_layers_shared(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), int(field), 0, _p(v, f64p))
        return v

    def layers_shared_set(self, field, values):
        v = np.ascontiguousarray(values, np.float64)
        self._lib.orc_layers_shared(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), int(field), 1, _p(v, f64p))

    def l_doubles(self):
        return 14          # the rank engine always packs the full record (oracle_capi.cpp: kL)

    def iter_begin(self):
        self.o.phaseA()
        self._lib.orc_halo_packA(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), len(self.sendShared),
                                 _p(self.sendShared, i32p), self.ptr["sendA"])
        if self.layers:
            self._lib.orc_halo_packL(self.o._h, _p(self.sharedLocal, i32p), len(self.sendShared), _p(self.sendShared, i32p), self.ptrL[0])

    def iter_interior(self):
        pass   # the oracle has no exchange-independent stage

    def iter_mid(self):
        self._lib.orc_halo_combineA(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), _p(self.combOffsets, i32p),
                                    _p(self.combSlots, i32p), self.ptr["recvA"])
        if self.layers:
            self._lib.orc_halo_combineL(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), _p(self.combOffsets, i32p),
                                        _p(self.combSlots, i32p), self.ptrL[1])
        self.o.phaseB()
        self._lib.orc_halo_packF(self.o._h, _p(self.sharedLocal, i32p), len(self.sendShared), _p(self.sendShared, i32p),
                                 self.ptr["sendF"])

    def iter_ahead(self):
        pass

    def set_stats_history(self, ptr, capacity):
        self._hist = (C.cast(ptr, f64p), int(capacity), 0) if ptr else None

    def iter_end(self):
        self._lib.orc_halo_orF(self.o._h, len(self.sharedLocal), _p(self.sharedLocal, i32p), _p(self.combOffsets, i32p),
                               _p(self.combSlots, i32p), self.ptr["recvF"])
        self.o.phaseC()
        self._lib.orc_local_stats(self.o._h, self.ptr["localStats"])
        if getattr(self, "_hist", None):
            p, cap, n = self._hist
            p[2 * (n % cap)], p[2 * (n % cap) + 1] = self.ptr["localStats"][0], self.ptr["localStats"][1]
            self._hist = (p, cap, n + 1)
        self.o.commit()

    def get_points(self):
        return self.o.points()


class MultiOracle:
    """Several Domains in lock-step with syncPointList semantics (the reference under mpirun)."""

    def __init__(self, oracles, sharedOff, sharedDomain, sharedLocal):
        self._lib = lib()
        self._oracles = oracles
        arr = (C.c_void_p * len(oracles))(*[o._h for o in oracles])
        self._h = self._lib.orc_multi_create(len(oracles), arr)
        so = np.ascontiguousarray(sharedOff, np.int32)
        sd = np.ascontiguousarray(sharedDomain, np.int32)
        sl = np.ascontiguousarray(sharedLocal, np.int32)
        self._lib.orc_multi_set_shared(self._h, len(so) - 1, _p(so, i32p), _p(sd, i32p), _p(sl, i32p))

    def set_sync_variant(self, variant):
        """"master" (default) / "own": the syncTools::syncPointList model (smooth_oracle.cpp, MultiDomain::syncVariant)"""
        self._lib.orc_multi_set_sync_variant(self._h, SYNC_VARIANTS[variant])

    def setup_layers(self, patch_arrays_per_domain, layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio, minLayers,
                     maxLayers):
        """boundary layer treatment under -parallel (set-up SM.C:2215-2221 with its syncPointList calls);
        patch_arrays_per_domain[d] = (start, size, kind, isLayer) of domain d; returns doLayerTreatment"""
        npd = np.array([len(a[0]) for a in patch_arrays_per_domain], np.int32)
        cat = lambda k, t: np.ascontiguousarray(np.concatenate([np.asarray(a[k]) for a in patch_arrays_per_domain]), t)
        st, sz, kd, il = cat(0, np.int32), cat(1, np.int32), cat(2, np.int32), cat(3, np.uint8)
        self._lib.orc_multi_setup_layers(self._h, _p(npd, i32p), _p(st, i32p), _p(sz, i32p), _p(kd, i32p), _p(il, u8p),
                                         layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio, minLayers, maxLayers)
        return bool(self._lib.orc_layers_enabled(self._oracles[0]._h))

    def setup_boundary(self, patch_arrays_per_domain, layer, initEdges, targetEdges, surf, internalSmoothingBlendingFraction=0.0):
        """boundary point smoothing under -parallel (set-up SM.C:2080-2253 with its reductions and syncPointList calls);
        patch_arrays_per_domain[d] = (start, size, kind, isLayer, isSmoothing) of domain d; layer as Oracle.setup_boundary"""
        npd = np.array([len(a[0]) for a in patch_arrays_per_domain], np.int32)
        cat = lambda k, t: np.ascontiguousarray(np.concatenate([np.asarray(a[k]) for a in patch_arrays_per_domain]), t)
        st, sz, kd, il, ism = cat(0, np.int32), cat(1, np.int32), cat(2, np.int32), cat(3, np.uint8), cat(4, np.uint8)
        def pe(m):
            if m is None:
                return np.zeros((0, 3), np.float64), np.zeros((0, 2), np.int32)
            return np.ascontiguousarray(m[0], np.float64).reshape(-1, 3), np.ascontiguousarray(m[1], np.int32)
        ip, ie = pe(initEdges); tp, te = pe(targetEdges); sp, stri = pe(surf)
        rc = self._lib.orc_multi_setup_boundary(self._h, _p(npd, i32p), _p(st, i32p), _p(sz, i32p), _p(kd, i32p), _p(il, u8p), _p(ism, u8p),
                                                float(layer[0]), float(layer[1]), float(layer[2]), int(layer[3]), int(layer[4]),
                                                len(ip), _p(ip, f64p), len(ie), _p(ie, i32p), len(tp), _p(tp, f64p), len(te), _p(te, i32p),
                                                len(sp), _p(sp, f64p), len(stri), _p(stri, i32p), float(internalSmoothingBlendingFraction))
        if rc < 0:
            raise RuntimeError("; ".join(o.error() for o in self._oracles if o.error()))
        return bool(rc)

    def iterate(self, nIters, relTol=0.02):
        res = np.zeros(max(nIters, 1), np.float64)
        frz = np.zeros(max(nIters, 1), np.int32)
        n = self._lib.orc_multi_iterate(self._h, nIters, relTol, _p(res, f64p), _p(frz, i32p))
        if n < 0:
            raise RuntimeError("oracle multi-domain iterate failed")
        return n, res[:n].copy(), frz[:n].astype(np.int64)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orc_multi_destroy(self._h)
            self._h = None


ACOS_VARIANTS = {"glibc": 0, "device": 1}


def set_acos_variant(variant):
    """process-wide: "glibc" (default: std::acos, the reference's arithmetic) / "device" (the algorithm the kernels evaluate,
    smoothmesh_amd/csrc/smacos.hpp: a fixed sequence of IEEE operations, bit-identical on CPU and GPU -- with it the engine's angle
    fields are compared bit for bit).  Returns the previous variant's name."""
    l = lib()
    prev = {v: k for k, v in ACOS_VARIANTS.items()}[l.orc_get_acos_variant()]
    l.orc_set_acos_variant(ACOS_VARIANTS[variant])
    return prev


def acos(x, variant="device"):
    return float(lib().orc_acos(float(x), ACOS_VARIANTS[variant]))


def acos_census_window(ulps=4):
    """window of the near-tie classes of acos_census (the engine's SMGPU_NEARTIE_ULPS)"""
    lib().orc_acos_census_window(int(ulps))


def acos_census(enable=None):
    """enable=True: reset and start counting the threshold comparisons of angles (SM.C:923, 1367, 1391-1394, 1421-1424); False: stop.
    Returns {"comparisons", "equal" (both sides the same bits: the same function of the same inputs), "within_8ulp" (sides 1 .. 8 ulp
    apart: what a last-bit difference between two acos implementations could flip), "min_ulp" (over the unequal pairs)}."""
    l = lib()
    if enable is not None:
        l.orc_acos_census_enable(1 if enable else 0)
    out = (C.c_longlong * 4)()
    l.orc_acos_census(out)
    near = (C.c_longlong * 3)()
    l.orc_acos_census_near(near)
    # near = the engine's near-tie census (include/smgpu.h) counted on the oracle's side: sides 1 .. window ulp apart, by comparison
    return {"comparisons": int(out[0]), "equal": int(out[1]), "within_8ulp": int(out[2]), "min_ulp": None if out[3] < 0 else int(out[3]),
            "near": {"edge_angle": int(near[0]), "good_range": int(near[1]), "walk": int(near[2])}}


def edge_strings(nPoints, edges):
    """findEdgeMeshStrings BPS.C:557-587 -> (string index per edge, number of strings)"""
    e = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    out = np.empty(len(e), np.int32)
    n = lib().orc_edge_strings(int(nPoints), len(e), _p(e, i32p), _p(out, i32p))
    return out, n


def edgeEdgeAngle(c, p1, p2):
    a, pa = _v(c); b, pb = _v(p1); d, pd = _v(p2)
    return lib().orc_edgeEdgeAngle(pa, pb, pd)


def calcEdgeCenterEdgeAngle(p0, cC, p1):
    a, pa = _v(p0); b, pb = _v(cC); d, pd = _v(p1)
    return lib().orc_calcEdgeCenterEdgeAngle(pa, pb, pd)


def isCloserPoint(a, b):
    x, px = _v(a); y, py = _v(b)
    return bool(lib().orc_isCloserPoint(px, py))
