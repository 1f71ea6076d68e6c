"""CPU tests of the product's host logic: the C-ABI library loads and exports every symbol the header
declares, the host-side addressing build agrees with the oracle's independent build (OpenFOAM orderings),
decomposition matches the direct sub-domain generator, the engine fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from smoothmesh_amd import _ffi
    hdr = open(os.path.join(ROOT, "include", "smgpu.h")).read()
    declared = set(re.findall(r"\b(smgpu_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 20
    l = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(l, name), f"libsmgpu.so does not export {name}"
    assert declared == set(_ffi.SYMBOLS), (declared ^ set(_ffi.SYMBOLS))
    assert b"gfx950" in _ffi.lib().smgpu_version()
    # include/smhost.h (polyMesh I/O, mesh generator) against libsmhost.so
    from smoothmesh_amd import polymesh
    hdr2 = open(os.path.join(ROOT, "include", "smhost.h")).read()
    declared2 = set(re.findall(r"\b(smhost_[a-z_]+)\s*\(", hdr2))
    assert len(declared2) >= 10
    l2 = ctypes.CDLL(polymesh.LIB_PATH)
    for name in declared2:
        assert hasattr(l2, name), f"libsmhost.so does not export {name}"


@pytest.mark.parametrize("dims,jit", [((5, 4, 3), 0.2), ((2, 2, 2), 0.0), ((7, 1, 2), 0.1)])
def test_host_topology_matches_oracle_addressing(oracle_lib, dims, jit):
    from smoothmesh_amd.engine import HostTopology
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(*dims, jitter=jit)
    o = oracle_lib.Oracle(m)
    t = HostTopology(m)
    for kind in ("pointCells", "pointPoints", "pointEdges", "pointFaces", "edgeFaces", "edgeCells", "edges"):
        a, b = o.addressing(kind), t.addressing(kind)
        if a[0] is not None:
            assert np.array_equal(a[0], b[0]), kind
        assert np.array_equal(a[1], b[1]), kind
    # OpenFOAM orderings the results depend on
    off, pp = t.addressing("pointPoints")
    for p in range(m.nPoints):
        row = pp[off[p]:off[p + 1]]
        assert np.all(np.diff(row) > 0)                    # ascending neighbour id
    _, edges = t.addressing("edges")
    assert np.all(edges[:, 0] < edges[:, 1])
    key = edges[:, 0].astype(np.int64) * m.nPoints + edges[:, 1]
    assert np.all(np.diff(key) > 0)                        # upper-triangular edge order
    # prev/next vertex tables against the face loops
    offf, pf = t.addressing("pointFaces")
    _, prv = t.addressing("pointFacePrev")
    _, nxt = t.addressing("pointFaceNext")
    for p in range(0, m.nPoints, 7):
        for k in range(offf[p], offf[p + 1]):
            f = pf[k]
            loop = list(m.facePoints[m.faceOffsets[f]:m.faceOffsets[f + 1]])
            i = loop.index(p)
            assert prv[k] == loop[i - 1] and nxt[k] == loop[(i + 1) % len(loop)]


def test_topology_rejects_bad_input():
    from smoothmesh_amd.engine import HostTopology, SmgpuError
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(2)
    m.owner = m.owner.copy(); m.owner[3] = 99
    with pytest.raises(SmgpuError):
        HostTopology(m)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smoothmesh_amd import SmoothEngine, SmgpuError
    from smoothmesh_amd.meshgen import hex_block
    with pytest.raises(SmgpuError, match="no HIP device"):
        SmoothEngine(hex_block(2))


def test_find_internal_points_and_empty_patch():
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(3)
    ip = m.find_internal_points()
    assert ip.sum() == 8 and ip.dtype == np.uint8          # 2^3 interior lattice points
    m.patches[0].type = "empty"
    with pytest.raises(ValueError, match="empty"):
        m.find_internal_points()                           # SM.C:61-66
    m.patches[0].type = "processor"                        # processor-patch points stay internal (SM.C:57)
    assert m.find_internal_points().sum() == 8 + 4


@pytest.mark.parametrize("grid,nLocal", [((2, 1, 1), (3, 4, 2)), ((2, 2, 2), (2, 3, 2)), ((1, 3, 1), (2, 2, 2))])
def test_decompose_matches_direct_subdomains(grid, nLocal):
    from smoothmesh_amd.decompose import decompose, grid_partition, shared_point_table
    from smoothmesh_amd.meshgen import hex_block, hex_subdomain
    world = grid[0] * grid[1] * grid[2]
    lengths = tuple(float(g) for g in grid)
    g = hex_block(nLocal[0] * grid[0], nLocal[1] * grid[1], nLocal[2] * grid[2], lengths=lengths, jitter=0.2, seed=5)
    subs = decompose(g, grid_partition(g, grid), world)
    for r, s in enumerate(subs):
        d = hex_subdomain(nLocal, grid, r, jitter=0.2, seed=5)
        for f in ("points", "faceOffsets", "facePoints", "owner", "neighbour"):
            assert np.array_equal(getattr(s.mesh, f), getattr(d.mesh, f)), (r, f)
        assert np.array_equal(s.pointProcAddressing, d.pointProcAddressing)
        assert [(p.name, p.type, p.nFaces, p.startFace, p.neighbProcNo) for p in s.mesh.patches] == \
               [(p.name, p.type, p.nFaces, p.startFace, p.neighbProcNo) for p in d.mesh.patches]
        assert np.array_equal(g.points[s.pointProcAddressing], s.mesh.points)
    off, dom, loc = shared_point_table(subs)
    assert np.all(np.diff(off) >= 2)


@pytest.mark.parametrize("N,grid", [(8, (2, 1, 1)), (9, (2, 2, 1)), (12, (2, 2, 2)), (11, (3, 2, 1)), (13, (3, 3, 3))])
def test_cavity_subdomain_matches_decompose_of_the_global_mesh(N, grid):
    """BASELINE configs[4]: the rank-local polyhedral generator (no global mesh in any process) gives, array for array,
    what cutting the global castellated mesh in decomposePar layout gives -- hanging-node faces on processor patches,
    reversed processor faces, points on the cavity wall that only another rank's cell touches (not jittered)"""
    from smoothmesh_amd.decompose import decompose, shared_point_table
    from smoothmesh_amd.polymesh import cavity_mesh, cavity_partition, cavity_subdomain
    g = cavity_mesh(N, jitter=0.2, seed=7)
    cellRank, lattice = cavity_partition(N, grid)
    world = grid[0] * grid[1] * grid[2]
    ref = decompose(g, cellRank, world)
    subs = [cavity_subdomain(N, grid, r, jitter=0.2, seed=7) for r in range(world)]
    for r, (a, b) in enumerate(zip(ref, subs)):
        for f in ("points", "faceOffsets", "facePoints", "owner", "neighbour"):
            assert np.array_equal(getattr(a.mesh, f), getattr(b.mesh, f)), (r, f)
        assert a.mesh.nCells == b.mesh.nCells
        key = lambda m: [(p.name, p.type, p.nFaces, p.startFace, p.myProcNo, p.neighbProcNo) for p in m.patches]
        assert key(a.mesh) == key(b.mesh)
        assert np.array_equal(lattice[a.pointProcAddressing], b.pointProcAddressing)
    # polygonal faces (hanging mid-edge points) do lie on processor patches, and some points have >= 3 sharers
    if world >= 4:
        off, dom, loc = shared_point_table(subs)
        assert np.diff(off).max() >= 3
    sizes = np.concatenate([np.diff(s.mesh.faceOffsets)[p.startFace:p.startFace + p.nFaces] for s in subs for p in s.mesh.patches
                            if p.type == "processor"])
    assert sizes.max() > 4


def test_halo_tables_are_symmetric():
    from smoothmesh_amd.halo import HaloTables
    from smoothmesh_amd.meshgen import hex_subdomain
    grid = (2, 2, 1)
    subs = [hex_subdomain((3, 3, 2), grid, r) for r in range(4)]
    cands = [s.processor_patch_point_lists() for s in subs]
    tabs = [HaloTables(r, subs[r].pointProcAddressing, cands) for r in range(4)]
    for a in range(4):
        for b in range(4):
            assert tabs[a].counts[b] == tabs[b].counts[a]
    # the centre line is shared by all 4 ranks
    assert max(np.diff(tabs[0].combOffsets)) == 4
    for t in tabs:
        assert (t.combSlots == -1).sum() == len(t.sharedLocal)
        assert np.array_equal(np.sort(t.combSlots[t.combSlots >= 0]), np.arange(t.nRecv))


def test_multi_domain_oracle_one_domain_equals_single(oracle_lib):
    """The multi-domain emulation with the mesh NOT split must be the single-domain loop."""
    from smoothmesh_amd import default_params
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(5, jitter=0.3, seed=3)
    a = oracle_lib.Oracle(m); b = oracle_lib.Oracle(m)
    p = default_params(a.mesh_stats()[0])
    a.set_params(p); b.set_params(p)
    mo = oracle_lib.MultiOracle([b], np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    ra = a.iterate(5, 0.0); rb = mo.iterate(5, 0.0)
    assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
    assert np.array_equal(a.points(), b.points())


def test_rccl_unique_id_marshalling_keeps_all_128_bytes():
    """the ncclUniqueId travels through the process group as bytes: NUL bytes inside it must survive (a c_char array field
    reads back truncated at the first NUL -- the bootstrap address then points nowhere)"""
    import ctypes as C
    from smoothmesh_amd.rccl_direct import _UniqueId
    assert C.sizeof(_UniqueId) == 128
    u = _UniqueId()
    raw = bytes((7 * i + 3) % 256 if i % 5 else 0 for i in range(128))
    C.memmove(C.byref(u), raw, 128)
    blob = C.string_at(C.byref(u), 128)
    assert blob == raw
    v = _UniqueId()
    C.memmove(C.byref(v), blob, 128)
    assert bytes(v.internal) == raw


def test_connected_components_numbering():
    """decompose._connected_components (numpy only) against a plain union-find: same partition, components numbered in the order of
    their smallest node"""
    import numpy as np
    from smoothmesh_amd.decompose import _connected_components
    rng = np.random.default_rng(5)
    for _ in range(100):
        n = int(rng.integers(1, 50))
        e = int(rng.integers(0, 70))
        ea, eb = rng.integers(0, n, e), rng.integers(0, n, e)
        parent = list(range(n))
        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for a, b in zip(ea, eb):
            ra, rb = find(int(a)), find(int(b))
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
        roots = [find(i) for i in range(n)]
        order = {r: k for k, r in enumerate(sorted(set(roots)))}
        assert np.array_equal(_connected_components(n, ea, eb), [order[r] for r in roots])
