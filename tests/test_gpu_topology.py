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


@pytest.mark.parametrize("which", range(6))
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


@pytest.mark.parametrize("which", [0, 3, 4, 5])
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
            assert (which_tables + " tiles: tables on the device" in log) == (dev == "1"), log
        sums = e.debug_tile_checksums()
        e.set_params(default_params(e.mesh_stats()[0]))
        e.iterate(3, 0.0)
        got.append((sums, e.get_points().copy()))
        e.close()
    for other in got[1:]:
        assert got[0][0] == other[0], (name, [i for i, (a, b) in enumerate(zip(got[0][0], other[0])) if a != b])
        assert any(got[0][0]) and np.array_equal(got[0][1], other[1])
