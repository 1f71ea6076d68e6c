"""ctypes binding of include/smgpu.h (the drop-in boundary).  No torch types cross it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOOTHMESH_SMGPU_LIB", os.path.join(_HERE, "csrc", "libsmgpu.so"))   # override: experimental builds

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)


class MeshDesc(C.Structure):
    _fields_ = [
        ("nPoints", C.c_int32), ("nCells", C.c_int32), ("nFaces", C.c_int32), ("nInternalFaces", C.c_int32),
        ("points", c_f64p), ("faceOffsets", c_i32p), ("facePoints", c_i32p), ("owner", c_i32p),
        ("neighbour", c_i32p), ("isInternalPoint", c_u8p), ("isSmoothingSurfacePoint", c_u8p),
        ("device", C.c_int32), ("stream", C.c_void_p), ("useCallerStream", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("maxStepLength", C.c_double), ("relStepFrac", C.c_double), ("minEdgeLength", C.c_double),
        ("totalMinFreeze", C.c_int32), ("edgeAngleConstraint", C.c_int32), ("faceAngleConstraint", C.c_int32),
        ("minAngle", C.c_double), ("maxAngle", C.c_double),
    ]


class IterStats(C.Structure):
    _fields_ = [("residual", C.c_double), ("nFrozenPoints", C.c_int32), ("nNearTies", C.c_int32)]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "nPoints", "nCells", "nFaces", "nInternalFaces", "nEdges", "nnzFacePoints", "nnzPointCells",
        "nnzPointPoints", "nnzPointFaces", "nnzEdgeFaces", "nnzEdgeCells", "nnzCellFaces", "deviceBytes")]


MAX_KERNELS = 24


class Counters(C.Structure):
    _fields_ = [
        ("nKernels", C.c_int32), ("name", C.c_char_p * MAX_KERNELS), ("ms", C.c_double * MAX_KERNELS),
        ("launches", C.c_int64 * MAX_KERNELS), ("algoBytesPerLaunch", C.c_int64 * MAX_KERNELS),
        ("algoF64OpsPerLaunch", C.c_int64 * MAX_KERNELS),
    ]


class HaloDesc(C.Structure):
    _fields_ = [
        ("nShared", C.c_int32), ("sharedLocal", c_i32p), ("nSend", C.c_int32), ("sendShared", c_i32p),
        ("nRecv", C.c_int32), ("combOffsets", c_i32p), ("combSlots", c_i32p),
        ("sendA", C.c_void_p), ("recvA", C.c_void_p), ("sendF", C.c_void_p), ("recvF", C.c_void_p),
        ("localStats", C.c_void_p), ("sendL", C.c_void_p), ("recvL", C.c_void_p),
        ("useExchangeStream", C.c_int32), ("exchangeStream", C.c_void_p),
    ]


class PushDesc(C.Structure):
    _fields_ = [
        ("nPeers", C.c_int32), ("peerCount", c_i32p), ("remoteBase", c_i32p), ("myIndexAtPeer", c_i32p),
        ("peerRecvA", C.POINTER(C.c_void_p)), ("peerRecvL", C.POINTER(C.c_void_p)), ("peerRecvF", C.POINTER(C.c_void_p)),
        ("peerFlags", C.POINTER(C.c_void_p)), ("localFlags", C.c_void_p),
    ]


class LayerDesc(C.Structure):
    _fields_ = [
        ("nPatches", C.c_int32), ("patchStart", c_i32p), ("patchSize", c_i32p), ("patchKind", c_u8p), ("isLayerPatch", c_u8p),
        ("layerMaxBlendingFraction", C.c_double), ("layerEdgeLength", C.c_double), ("layerExpansionRatio", C.c_double),
        ("minLayers", C.c_int32), ("maxLayers", C.c_int32),
    ]


class BoundaryDesc(C.Structure):
    _fields_ = [
        ("nPatches", C.c_int32), ("patchStart", c_i32p), ("patchSize", c_i32p), ("patchKind", c_u8p), ("isSmoothingPatch", c_u8p),
        ("nInitEdgePoints", C.c_int32), ("initEdgePoints", c_f64p), ("nInitEdges", C.c_int32), ("initEdges", c_i32p),
        ("nTargetEdgePoints", C.c_int32), ("targetEdgePoints", c_f64p), ("nTargetEdges", C.c_int32), ("targetEdges", c_i32p),
        ("nSurfacePoints", C.c_int32), ("surfacePoints", c_f64p), ("nSurfaceTriangles", C.c_int32), ("surfaceTriangles", c_i32p),
        ("isCornerPointIO", c_i32p), ("isFeatureEdgePointIO", c_i32p),
        ("distanceTolerance", C.c_double), ("internalSmoothingBlendingFraction", C.c_double),
    ]


class BoundaryInfo(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("nCornerPoints", C.c_int32), ("nFeatureEdgePoints", C.c_int32),
                ("nSmoothingSurfacePoints", C.c_int32), ("nFrozenSurfacePoints", C.c_int32), ("nTargetEdgeStrings", C.c_int32)]


# every symbol include/smgpu.h declares: (restype, argtypes)
SYMBOLS = {
    "smgpu_last_error": (C.c_char_p, []),
    "smgpu_version": (C.c_char_p, []),
    "smgpu_create": (C.c_int, [C.POINTER(MeshDesc), C.POINTER(C.c_void_p)]),
    "smgpu_destroy": (C.c_int, [C.c_void_p]),
    "smgpu_get_sizes": (C.c_int, [C.c_void_p, C.POINTER(Sizes)]),
    "smgpu_mesh_stats": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "smgpu_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "smgpu_set_foam_variant": (C.c_int, [C.c_void_p, C.c_int32]),
    "smgpu_set_sync_variant": (C.c_int, [C.c_void_p, C.c_int32]),
    "smgpu_halo_set_push": (C.c_int, [C.c_void_p, C.POINTER(PushDesc)]),
    "smgpu_push_alloc": (C.c_int, [C.c_int32, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]),
    "smgpu_push_open": (C.c_int, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "smgpu_push_close": (C.c_int, [C.c_void_p]),
    "smgpu_push_free": (C.c_int, [C.c_void_p]),
    "smgpu_set_device_share": (C.c_int, [C.c_void_p, C.c_int32]),
    "smgpu_debug_walk_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "smgpu_debug_halo_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "smgpu_debug_selftest_fpexact": (C.c_int, [C.c_int32, C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]),
    "smgpu_iterate": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.POINTER(IterStats), c_i32p]),
    "smgpu_get_points": (C.c_int, [C.c_void_p, c_f64p]),
    "smgpu_check_error": (C.c_int, [C.c_void_p]),
    "smgpu_get_near_ties": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "smgpu_set_points": (C.c_int, [C.c_void_p, c_f64p]),
    "smgpu_enable_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "smgpu_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "smgpu_reset_counters": (C.c_int, [C.c_void_p]),
    "smgpu_halo_configure": (C.c_int, [C.c_void_p, C.POINTER(HaloDesc)]),
    "smgpu_iter_begin": (C.c_int, [C.c_void_p]),
    "smgpu_iter_interior": (C.c_int, [C.c_void_p]),
    "smgpu_iter_mid": (C.c_int, [C.c_void_p]),
    "smgpu_iter_ahead": (C.c_int, [C.c_void_p]),
    "smgpu_halo_set_stats_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "smgpu_set_layers": (C.c_int, [C.c_void_p, C.POINTER(LayerDesc), c_i32p]),
    "smgpu_layers_begin": (C.c_int, [C.c_void_p, C.POINTER(LayerDesc), c_i32p, c_i32p]),
    "smgpu_layers_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "smgpu_layers_shared": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_f64p]),
    "smgpu_set_boundary_smoothing": (C.c_int, [C.c_void_p, C.POINTER(BoundaryDesc), C.POINTER(BoundaryInfo)]),
    "smgpu_boundary_stats": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "smgpu_boundary_begin": (C.c_int, [C.c_void_p, C.POINTER(BoundaryDesc), C.c_double, C.c_double, C.POINTER(BoundaryInfo)]),
    "smgpu_boundary_step": (C.c_int, [C.c_void_p, C.c_int32]),
    "smgpu_boundary_shared": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_f64p]),
    "smgpu_get_boundary_classification": (C.c_int, [C.c_void_p, c_i32p, c_i32p]),
    "smgpu_debug_edge_strings": (C.c_int, [C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p]),
    "smgpu_debug_find_line": (C.c_int, [C.c_void_p, C.c_int32, c_f64p, c_f64p, c_i32p]),
    "smgpu_halo_l_doubles": (C.c_int, [C.c_void_p, c_i32p]),
    "smgpu_halo_set_exchange_stream": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "smgpu_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "smgpu_iter_end": (C.c_int, [C.c_void_p]),
    "smgpu_debug_get_field": (C.c_int, [C.c_void_p, C.c_char_p, c_f64p, C.POINTER(C.c_int64)]),
    "smgpu_debug_get_addressing": (C.c_int, [C.c_void_p, C.c_char_p, c_i32p, c_i32p, C.POINTER(C.c_int64)]),
    "smgpu_debug_propose": (C.c_int, [C.c_void_p]),
    "smgpu_topology_create": (C.c_int, [C.POINTER(MeshDesc), C.POINTER(C.c_void_p)]),
    "smgpu_topology_get": (C.c_int, [C.c_void_p, C.c_char_p, c_i32p, c_i32p, C.POINTER(C.c_int64)]),
    "smgpu_topology_num_edges": (C.c_int, [C.c_void_p, c_i32p]),
    "smgpu_topology_checksums": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "smgpu_debug_addressing_checksums": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "smgpu_debug_tile_checksums": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "smgpu_topology_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


def lib():
    """Load csrc/libsmgpu.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C smoothmesh_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same soname as /opt/rocm's).
        # Whichever is loaded first serves both; if ours came first torch would later fail with "No HIP GPUs
        # are available", so let torch (when present) load its runtime before libsmgpu.so is opened.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
