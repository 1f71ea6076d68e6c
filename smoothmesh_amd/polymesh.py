"""OpenFOAM case-directory I/O through the C++ host library (include/smhost.h): the step either side
of the loop -- createMesh.H (src/smoothMesh.C:1814-1818) and mesh.write() (SM.C:2416-2431)."""
import ctypes as C
import os

import numpy as np

from .mesh import PolyMesh, Patch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOOTHMESH_SMHOST_LIB", os.path.join(_HERE, "csrc", "libsmhost.so"))   # override: sanitizer builds
_lib = None
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C smoothmesh_amd/csrc`")
        l = C.CDLL(LIB_PATH)
        l.smhost_last_error.restype = C.c_char_p
        l.smhost_read_polymesh.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        l.smhost_mesh_free.argtypes = [C.c_void_p]
        l.smhost_mesh_sizes.argtypes = [C.c_void_p] + [i32p] * 5 + [C.POINTER(C.c_int64)]
        l.smhost_mesh_copy.argtypes = [C.c_void_p, f64p, i32p, i32p, i32p, i32p]
        l.smhost_mesh_patch.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, i32p, i32p, i32p, i32p]
        l.smhost_write_polymesh.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, f64p, C.c_int32, i32p, i32p, i32p, C.c_int32, i32p,
                                            C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i32p, i32p, i32p, i32p,
                                            C.c_int32, C.c_int32]
        l.smhost_write_points.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, f64p, C.c_int32, C.c_int32]
        l.smhost_set_write_compression.argtypes = [C.c_int32]
        l.smhost_read_label_list.argtypes = [C.c_char_p, i32p, C.POINTER(C.c_int64)]
        l.smhost_read_obj.argtypes = [C.c_char_p, C.c_int32, f64p, C.POINTER(C.c_int64), i32p, C.POINTER(C.c_int64)]
        l.smhost_write_label_list.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, i32p, C.c_int32]
        l.smhost_gen_cavity_mesh.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_void_p)]
        l.smhost_gen_cavity_subdomain.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                                  C.c_int32, C.POINTER(C.c_void_p)]
        l.smhost_mesh_global_ids.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = l
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(lib().smhost_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(t)


def _take(h, global_ids=None) -> PolyMesh:
    """copy a smhost_mesh out and free it; global_ids: a list that receives (pointGlobal, cellGlobal)"""
    l = lib()
    try:
        n = [C.c_int32() for _ in range(5)]
        nnz = C.c_int64()
        l.smhost_mesh_sizes(h, *[C.byref(x) for x in n], C.byref(nnz))
        nP, nC, nF, nIF, nPatch = [x.value for x in n]
        pts = np.empty((nP, 3)); fo = np.empty(nF + 1, np.int32); fp = np.empty(nnz.value, np.int32)
        ow = np.empty(nF, np.int32); ne = np.empty(nIF, np.int32)
        l.smhost_mesh_copy(h, _p(pts, f64p), _p(fo, i32p), _p(fp, i32p), _p(ow, i32p), _p(ne, i32p))
        patches = []
        for i in range(nPatch):
            name = C.create_string_buffer(256); typ = C.create_string_buffer(64)
            a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            _check(l.smhost_mesh_patch(h, i, name, 256, typ, 64, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
            patches.append(Patch(name.value.decode(), typ.value.decode(), a.value, b.value,
                                 c.value if c.value >= 0 else None, d.value if d.value >= 0 else None))
        if global_ids is not None:
            pg, cg = np.empty(nP, np.int64), np.empty(nC, np.int64)
            _check(l.smhost_mesh_global_ids(h, _p(pg, C.POINTER(C.c_int64)), _p(cg, C.POINTER(C.c_int64))))
            global_ids.extend([pg, cg])
        return PolyMesh(pts, fo, fp, ow, ne, patches, nC)
    finally:
        l.smhost_mesh_free(h)


def read_polymesh(polyMeshDir, pointsDir=None) -> PolyMesh:
    h = C.c_void_p()
    _check(lib().smhost_read_polymesh(polyMeshDir.encode(), pointsDir.encode() if pointsDir else None, C.byref(h)))
    return _take(h)


def cavity_mesh(N, radius=0.25, shell=None, jitter=0.2, seed=12345) -> PolyMesh:
    """Castellated one-level octree mesh of the unit cube with a spherical cavity (polyhedral cells at the
    refinement interface) -- stands in for snappyHexMesh, see csrc/host/meshgen.cpp.  shell = half-width of
    the refined band around the sphere surface (default 2 coarse cells)."""
    if shell is None:
        shell = 2.0 / N
    h = C.c_void_p()
    _check(lib().smhost_gen_cavity_mesh(N, radius, shell, jitter, seed, C.byref(h)))
    return _take(h)


def cavity_box_of(N, P, i):
    """box of coarse cell index i along an axis cut into P boxes: floor(b N / P) <= i < floor((b + 1) N / P)"""
    i = np.asarray(i, np.int64)
    lo = (np.arange(P + 1, dtype=np.int64) * N) // P
    return (np.searchsorted(lo, i, side="right") - 1).astype(np.int64)


def cavity_subdomain(N, grid, rank, radius=0.25, shell=None, jitter=0.2, seed=12345, with_cells=False):
    """Sub-domain `rank` of cavity_mesh(N, ...) cut into grid[0] x grid[1] x grid[2] boxes of coarse cells, generated
    directly in decomposePar layout (csrc/host/meshgen.cpp): identical to decompose(cavity_mesh(N), cavity_partition)[rank]
    (tests/test_host_logic.py), but no process ever holds the global mesh -- the analogue of meshgen.hex_subdomain for
    BASELINE configs[4].  pointProcAddressing holds global LATTICE indices (unique per global point, ascending with the
    global point id), which is all the shared-point tables need."""
    from .decompose import SubDomain
    if shell is None:
        shell = 2.0 / N
    h = C.c_void_p()
    _check(lib().smhost_gen_cavity_subdomain(N, radius, shell, jitter, seed, grid[0], grid[1], grid[2], rank, C.byref(h)))
    ids = []
    mesh = _take(h, ids)
    return SubDomain(mesh, rank, grid[0] * grid[1] * grid[2], ids[0], ids[1] if with_cells else None)


def cavity_partition(N, grid, radius=0.25, shell=None):
    """(cellRank of every cell of cavity_mesh(N), global lattice index of every point) under cavity_subdomain's box rule"""
    if shell is None:
        shell = 2.0 / N
    h = C.c_void_p()
    _check(lib().smhost_gen_cavity_subdomain(N, radius, shell, 0.0, 0, 1, 1, 1, 0, C.byref(h)))
    ids = []
    _take(h, ids)
    coarse = ids[1] // 8
    ci, cj, ck = coarse % N, (coarse // N) % N, coarse // (N * N)
    r = cavity_box_of(N, grid[0], ci) + grid[0] * (cavity_box_of(N, grid[1], cj) + grid[1] * cavity_box_of(N, grid[2], ck))
    return r.astype(np.int32), ids[0]


def write_polymesh(polyMeshDir, mesh: PolyMesh, location="constant/polyMesh", binary=False, precision=17):
    names = (C.c_char_p * len(mesh.patches))(*[p.name.encode() for p in mesh.patches])
    types = (C.c_char_p * len(mesh.patches))(*[p.type.encode() for p in mesh.patches])
    nf = np.array([p.nFaces for p in mesh.patches], np.int32)
    st = np.array([p.startFace for p in mesh.patches], np.int32)
    my = np.array([-1 if p.myProcNo is None else p.myProcNo for p in mesh.patches], np.int32)
    nb = np.array([-1 if p.neighbProcNo is None else p.neighbProcNo for p in mesh.patches], np.int32)
    pts = np.ascontiguousarray(mesh.points)
    _check(lib().smhost_write_polymesh(polyMeshDir.encode(), location.encode(), mesh.nPoints, _p(pts, f64p), mesh.nFaces,
                                       _p(mesh.faceOffsets, i32p), _p(mesh.facePoints, i32p), _p(mesh.owner, i32p),
                                       mesh.nInternalFaces, _p(mesh.neighbour, i32p), mesh.nCells, len(mesh.patches), names, types,
                                       _p(nf, i32p), _p(st, i32p), _p(my, i32p), _p(nb, i32p), int(binary), precision))


def set_write_compression(on):
    """controlDict writeCompression: files written afterwards become <file>.gz (reading accepts both always)"""
    _check(lib().smhost_set_write_compression(int(bool(on))))


def write_points(polyMeshDir, points, location, binary=False, precision=10):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    _check(lib().smhost_write_points(polyMeshDir.encode(), location.encode(), pts.shape[0], _p(pts, f64p), int(binary), precision))


def read_label_list(path):
    n = C.c_int64(-1)
    _check(lib().smhost_read_label_list(path.encode(), None, C.byref(n)))
    out = np.empty(n.value, np.int32)
    _check(lib().smhost_read_label_list(path.encode(), _p(out, i32p), C.byref(n)))
    return out


def read_obj(path, kind):
    """the front-end's OBJ reader (csrc/host/polymesh_io.cpp): kind "surface" -> (points, triangles), "edges" -> (points, edges)"""
    k = {"surface": 0, "edges": 1}[kind]
    nP, nE = C.c_int64(0), C.c_int64(0)
    _check(lib().smhost_read_obj(path.encode(), k, None, C.byref(nP), None, C.byref(nE)))
    pts, el = np.empty((nP.value, 3), np.float64), np.empty((nE.value, 3 if k == 0 else 2), np.int32)
    _check(lib().smhost_read_obj(path.encode(), k, _p(pts, f64p), C.byref(nP), _p(el, i32p), C.byref(nE)))
    return pts, el


def write_label_list(path, values, location, obj, cls="labelList", binary=False):
    v = np.ascontiguousarray(values, dtype=np.int32)
    _check(lib().smhost_write_label_list(path.encode(), location.encode(), obj.encode(), cls.encode(), len(v), _p(v, i32p), int(binary)))


CONTROL_DICT = """FoamFile
{
    version     2.0;
    format      ascii;
    class       dictionary;
    object      controlDict;
}
application     smoothMesh;
startFrom       latestTime;
startTime       0;
stopAt          endTime;
endTime         1000;
deltaT          1;
writeControl    timeStep;
writeInterval   1;
writeFormat     %s;
writePrecision  %d;
writeCompression %s;
timeFormat      general;
timePrecision   6;
"""


def write_case(caseDir, mesh: PolyMesh, binary=False, precision=17, writeFormat="ascii", writePrecision=10, writeCompression=False):
    """A minimal OpenFOAM case: system/controlDict + constant/polyMesh (gzip-compressed with writeCompression)."""
    os.makedirs(os.path.join(caseDir, "system"), exist_ok=True)
    with open(os.path.join(caseDir, "system", "controlDict"), "w") as f:
        f.write(CONTROL_DICT % (writeFormat, writePrecision, "on" if writeCompression else "off"))
    set_write_compression(writeCompression)
    try:
        write_polymesh(os.path.join(caseDir, "constant", "polyMesh"), mesh, binary=binary, precision=precision)
    finally:
        set_write_compression(False)


def write_decomposed_case(caseDir, subs, binary=False, precision=17, **kw):
    """processorN/constant/polyMesh + pointProcAddressing, decomposePar layout."""
    os.makedirs(os.path.join(caseDir, "system"), exist_ok=True)
    with open(os.path.join(caseDir, "system", "controlDict"), "w") as f:
        f.write(CONTROL_DICT % (kw.get("writeFormat", "ascii"), kw.get("writePrecision", 10), "off"))
    for s in subs:
        d = os.path.join(caseDir, f"processor{s.rank}", "constant", "polyMesh")
        write_polymesh(d, s.mesh, binary=binary, precision=precision)
        write_label_list(os.path.join(d, "pointProcAddressing"), s.pointProcAddressing, "constant/polyMesh", "pointProcAddressing",
                         "labelIOList", binary)
