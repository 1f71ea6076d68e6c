"""smoothmesh_amd -- MI355X-native smoothing-iteration engine for OpenFOAM polyMesh (hot path of
tkeskita/smoothMesh, src/smoothMesh.C:2257-2437) behind a C-ABI (include/smgpu.h).

The package is a thin ctypes mirror of that C-ABI plus the host-side pieces either side of the path
(polyMesh containers, synthetic mesh generators, decomposition, halo driver).  All arithmetic of the
path runs in csrc/libsmgpu.so (hand-written HIP); there is no CPU fallback: loading fails loudly when
the library has not been built.
"""
from .mesh import PolyMesh, Patch  # noqa: F401
from .engine import BoundaryParams, LayerParams, SmoothEngine, SmoothParams, SmgpuError, default_params, patch_arrays  # noqa: F401
