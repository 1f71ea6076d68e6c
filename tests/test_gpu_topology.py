"""The addressing built on the device (csrc/topology_dev.hip: radix sorts of (row, value) keys + per-edge kernels; what an engine
holds) against the host build (csrc/topology.cpp, smgpu_topology_create) -- every array, byte for byte (FNV checksums in a fixed
order), on hex blocks, the castellated polyhedral mesh (hanging-node faces, 2:1 interfaces), a mesh with a baffle (edges with four
faces and two disconnected cell fans), a tetrahedral / prismatic mix, and meshes the device path must hand back to the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(mesh, monkeypatch=None):
    from smoothmesh_amd import SmoothEngine
    from smoothmesh_amd.engine import TOPO_ARRAYS, HostTopology
    e = SmoothEngine(mesh)
    dev = e.debug_addressing_checksums()
    e.close()
    ht = HostTopology(mesh)
    host = ht.checksums()
    ht.close()
    return TOPO_ARRAYS, dev, host


def _hedgehog():
    """80 tetrahedra around one centre point (a once subdivided icosahedron's triangles joined to the centre): the centre has 120
    faces and 42 neighbours -- more corners than the device build of the corner chains handles (it hands the smoothing tables back),
    and more neighbours than the common-cell pair masks exist for -- while every edge keeps few faces (the device addressing stands)"""
    from smoothmesh_amd.polymesh import Patch, PolyMesh
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    tri = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
           (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, float) / np.linalg.norm(x) for x in v]
    mid = {}
    def midpoint(a, b):
        k = (min(a, b), max(a, b))
        if k not in mid:
            m = v[a] + v[b]
            v.append(m / np.linalg.norm(m))
            mid[k] = len(v) - 1
        return mid[k]
    tris = []
    for a, b, c in tri:
        ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
        tris += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
    rng = np.random.default_rng(11)
    pts = np.array([np.zeros(3)] + [x * (1.0 + 0.1 * rng.random()) for x in v])      # point 0 = the centre
    tris = [tuple(i + 1 for i in tr) for tr in tris]
    for k, (a, b, c) in enumerate(tris):      # outward
        if np.dot(np.cross(pts[b] - pts[a], pts[c] - pts[a]), pts[a]) < 0:
            tris[k] = (a, c, b)
    by_edge = {}
    for ti, tr in enumerate(tris):
        for i in range(3):
            by_edge.setdefault((min(tr[i], tr[(i + 1) % 3]), max(tr[i], tr[(i + 1) % 3])), []).append(ti)
    internal = []
    for (i, j), (t1, t2) in by_edge.items():
        t1, t2 = min(t1, t2), max(t1, t2)
        n = np.cross(pts[i], pts[j])
        f = [0, i, j] if np.dot(n, pts[list(tris[t2])].mean(axis=0)) > 0 else [0, j, i]
        internal.append((t1, t2, f))
    internal.sort(key=lambda x: (x[0], x[1]))
    faces = [f for _, _, f in internal] + [list(tr) for tr in tris]
    owner = [a for a, _, _ in internal] + list(range(len(tris)))
    neighbour = [b for _, b, _ in internal]
    off = np.concatenate([[0], np.cumsum([len(f) for f in faces])]).astype(np.int32)
    return PolyMesh(points=pts, faceOffsets=off, facePoints=np.concatenate(faces).astype(np.int32), owner=np.array(owner, np.int32),
                    neighbour=np.array(neighbour, np.int32), patches=[Patch("wall", "wall", len(tris), len(internal))])


def _meshes():
    from smoothmesh_amd.meshgen import add_baffle, baffle_in_plane, hex_block
    from smoothmesh_amd.polymesh import cavity_mesh
    yield "hex 9x7x5", hex_block(9, 7, 5, jitter=0.2, seed=1)
    yield "hex 1x1x1", hex_block(1, 1, 1)
    yield "hex 40x3x2", hex_block(40, 3, 2, jitter=0.1, seed=2)
    yield "polyhedral cavity 16", cavity_mesh(16, jitter=0.2, seed=3)
    yield "polyhedral cavity 30", cavity_mesh(30, jitter=0.2, seed=4)
    lattice = hex_block(8, 8, 6)
    yield "hex with a baffle", add_baffle(hex_block(8, 8, 6, jitter=0.1, seed=5), baffle_in_plane(lattice, 0, 0.5))
    yield "tetrahedra around one point", _hedgehog()


@pytest.mark.parametrize("which", range(7))
def test_device_addressing_equals_the_host_build(which):
    name, mesh = list(_meshes())[which]
    names, dev, host = _both(mesh)
    bad = [n for n, a, b in zip(names, dev, host) if a != b]
    assert not bad, (name, bad)


def test_the_engine_really_built_it_on_the_device(capfd, monkeypatch):
    """SMGPU_VERBOSE=1 says where the addressing was built; SMGPU_DEVICE_TOPOLOGY=0 is the host build -- same tables"""
    from smoothmesh_amd import SmoothEngine
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(10, 9, 8, jitter=0.2, seed=7)
    monkeypatch.setenv("SMGPU_VERBOSE", "1")
    e = SmoothEngine(mesh)
    a = e.debug_addressing_checksums()
    e.close()
    assert "addressing on the device" in capfd.readouterr().err
    monkeypatch.setenv("SMGPU_DEVICE_TOPOLOGY", "0")
    e = SmoothEngine(mesh)
    b = e.debug_addressing_checksums()
    e.close()
    assert "addressing on the device" not in capfd.readouterr().err
    assert a == b


def test_meshes_the_device_path_hands_back(capfd, monkeypatch):
    """an edge with more than 16 faces (a fan of 20 wedge cells around one edge) goes to the host build -- and gives its tables"""
    from smoothmesh_amd import SmoothEngine
    from smoothmesh_amd.engine import HostTopology
    from smoothmesh_amd.mesh import Patch, PolyMesh
    n = 20
    pts = [[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]]
    for k in range(n):
        a = 2 * np.pi * k / n
        pts += [[np.cos(a), np.sin(a), 0.0], [np.cos(a), np.sin(a), 1.0]]
    pts = np.array(pts)
    lo = lambda k: 2 + 2 * (k % n)
    hi = lambda k: 3 + 2 * (k % n)
    faces, owner, neighbour = [], [], []
    for k in range(n):            # internal faces: the radial planes, owner k - 1 (mod n) ... ordered so that owner < neighbour
        c0, c1 = (k - 1) % n, k
        o, nb = min(c0, c1), max(c0, c1)
        f = [0, 1, hi(k), lo(k)]
        # the face normal has to point from the owner to the neighbour
        if o == c1:
            f = f[::-1]
        faces.append(f); owner.append(o); neighbour.append(nb)
    order = sorted(range(n), key=lambda i: (owner[i], neighbour[i]))
    faces = [faces[i] for i in order]; owner = [owner[i] for i in order]; neighbour = [neighbour[i] for i in order]
    for k in range(n):            # boundary: outer quad, bottom and top triangles of wedge k (between radial planes k and k + 1)
        faces += [[lo(k), lo(k + 1), hi(k + 1), hi(k)], [0, lo(k + 1), lo(k)], [1, hi(k), hi(k + 1)]]
        owner += [k, k, k]
    off = np.concatenate([[0], np.cumsum([len(f) for f in faces])]).astype(np.int32)
    mesh = PolyMesh(points=pts, faceOffsets=off, facePoints=np.concatenate(faces).astype(np.int32), owner=np.array(owner, np.int32),
                    neighbour=np.array(neighbour, np.int32), patches=[Patch("wall", "wall", 3 * n, n)])
    monkeypatch.setenv("SMGPU_VERBOSE", "1")
    e = SmoothEngine(mesh)
    a = e.debug_addressing_checksums()
    e.close()
    assert "handed the mesh back" in capfd.readouterr().err
    ht = HostTopology(mesh)
    assert a == ht.checksums()
    ht.close()


@pytest.mark.parametrize("which", [0, 3, 4, 5, 6])
def test_device_tile_tables_equal_the_host_build(which, monkeypatch, capfd):
    """the geometry, smoothing and edge tile tables built on the device from the host's tile boundaries (csrc/tiles_dev.hip) against the host
    build (SMGPU_DEVICE_TILES=0): every table the kernels read, byte for byte -- and the same smoothing result"""
    import numpy as np
    from smoothmesh_amd import SmoothEngine, default_params
    name, mesh = list(_meshes())[which]
    got = []
    monkeypatch.setenv("SMGPU_VERBOSE", "2")
    for dev in ("1", "0", "2"):      # 2: the device builds hand their tables back to the host while its lists still arrive
        monkeypatch.setenv("SMGPU_DEVICE_TILES", dev)
        e = SmoothEngine(mesh)
        log = capfd.readouterr().err
        for which_tables in ("geometry", "smoothing", "edge"):
            on_device = dev == "1" and not (which == 6 and which_tables == "smoothing")      # (the hedgehog's centre: 120 corners)
            assert (which_tables + " tiles: tables on the device" in log) == on_device, log
        sums = e.debug_tile_checksums()
        e.set_params(default_params(e.mesh_stats()[0]))
        e.iterate(3, 0.0)
        got.append((sums, e.get_points().copy()))
        e.close()
    for other in got[1:]:
        assert got[0][0] == other[0], (name, [i for i, (a, b) in enumerate(zip(got[0][0], other[0])) if a != b])
        assert any(got[0][0]) and np.array_equal(got[0][1], other[1])


@pytest.mark.parametrize("constraints", [False, True])
def test_tetrahedra_around_one_point_match_the_oracle(constraints):
    """the mesh whose centre point has 120 face corners and 42 neighbours (no pair masks, corner chains on the host): the engine's
    coordinates against the oracle's, bit for bit"""
    from oracle import oracle_ffi
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = _hedgehog()
    o = oracle_ffi.Oracle(mesh)
    e = SmoothEngine(mesh)
    p = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(p)
    e.set_params(p)
    n_o, res_o, frz_o = o.iterate(10, 0.0)
    n_g, res_g, frz_g = e.iterate(10, 0.0)
    assert n_o == n_g == 10 and np.array_equal(frz_o, frz_g)
    assert np.array_equal(e.get_points(), o.points())
    e.close()
