"""GPU parity of the optional boundary point smoothing (include/smgpu.h smgpu_set_boundary_smoothing; reference set-up
SM.C:2080-2253, per iteration SM.C:2266 + 2307-2357) against the CPU oracle, through the C-ABI."""
import os

import numpy as np
import pytest

from bnd_cases import boundary_inputs, make_pair, scale_about_centre, tangential_jitter
from conftest import rel_linf

pytestmark = pytest.mark.gpu


def _check_setup(o, e):
    f = o.boundary_fields()
    cio, fio = e.boundary_classification()
    assert np.array_equal(cio, f["isCornerPoint"].astype(np.int32))
    assert np.array_equal(fio, f["isFeatureEdgePoint"].astype(np.int32))
    info = e.boundary_info
    assert info["nCornerPoints"] == f["isCornerPoint"].sum() and info["nFeatureEdgePoints"] == f["isFeatureEdgePoint"].sum()
    assert info["nSmoothingSurfacePoints"] == f["isSmoothingSurfacePoint"].sum()
    assert info["nTargetEdgeStrings"] == len(np.unique(f["targetEdgeStrings"]))
    assert np.array_equal(e.debug_field("layerNormals").reshape(-1, 3), f["normals"])     # same operations: same bits


def _run_both(o, e, iters):
    n_o, res_o, frz_o = o.iterate(iters, 0.0)
    n_g, res_g, frz_g = e.iterate(iters, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    assert rel_linf(e.get_points(), o.points()) <= 1e-13
    return res_g, frz_g


@pytest.mark.parametrize("constraints", [False, True])
@pytest.mark.parametrize("warp", [None, 1.03])
def test_hex_block_boundary_smoothing(oracle_lib, constraints, warp):
    from smoothmesh_amd.meshgen import hex_block
    m = tangential_jitter(hex_block(12, 10, 9, jitter=0.25, seed=11), 0.02, seed=3)
    # non-cubic block: lengths stay 1, the generators below describe the same unit cube
    init, target, surf = boundary_inputs(7, 5, warp=None if warp is None else scale_about_centre(warp))
    o, e, prm, on = make_pair(m, oracle_lib, init, target, surf, constraints=constraints)
    assert on
    _check_setup(o, e)
    _run_both(o, e, 12)
    # the boundary really moves: different from a run with the boundary frozen
    from smoothmesh_amd import SmoothEngine
    e2 = SmoothEngine(m)
    e2.set_params(prm)
    e2.iterate(12, 0.0)
    assert rel_linf(e2.get_points(), e.get_points()) > 1e-4


def test_boundary_smoothing_with_layers_and_blending(oracle_lib):
    """layer treatment on one patch + boundary smoothing + internalSmoothingBlendingFraction (OBB.C:573-631)"""
    from smoothmesh_amd.meshgen import hex_block
    m = tangential_jitter(hex_block(10, jitter=0.2, seed=7), 0.02, seed=5)
    init, target, surf = boundary_inputs(10, 4)
    o, e, prm, on = make_pair(m, oracle_lib, init, None, surf, layerPatches=("xmin",), blend=0.5)
    assert on
    _check_setup(o, e)
    _run_both(o, e, 10)


def test_only_some_patches_smoothed(oracle_lib):
    """-smoothingPatches '(xmin zmax)': the other boundary points are frozen surface points (BPS.C:414-420), yet corners
    and feature edge points among them are still projected before the restore (BPS.C:876-900, SM.C:2384-2392)"""
    from smoothmesh_amd.meshgen import hex_block
    m = tangential_jitter(hex_block(9, jitter=0.2, seed=8), 0.02, seed=6)
    init, target, surf = boundary_inputs(9, 3, warp=scale_about_centre(1.02))
    o, e, prm, on = make_pair(m, oracle_lib, init, target, surf, constraints=True, smoothingPatches=("xmin", "zmax"))
    assert on
    _check_setup(o, e)
    _run_both(o, e, 8)


@pytest.mark.parametrize("constraints", [False, True])
def test_polyhedral_cavity_snapped_to_a_sphere(oracle_lib, constraints):
    """castellated cavity wall projected to the sphere: most first searches miss (the wall is farther from the target
    than minEdgeLength) and the search radius grows (BPS.C:919-930); rays hit a curved, finely triangulated surface"""
    from smoothmesh_amd.polymesh import cavity_mesh
    from smoothmesh_amd.surfgen import box_feature_edges, sphere_surface
    m = cavity_mesh(12)
    o, e, prm, on = make_pair(m, oracle_lib, box_feature_edges(12), None, sphere_surface(levels=4), constraints=constraints,
                              smoothingPatches=("cavity",))
    assert on
    _check_setup(o, e)
    _run_both(o, e, 15)
    wall = o.boundary_fields()["isSmoothingSurfacePoint"].astype(bool)
    r = np.linalg.norm(e.get_points()[wall] - 0.5, axis=1)
    if not constraints:               # with the angle constraints some wall points are frozen on their way
        assert r.min() > 0.23 and r.max() <= 0.25 + 1e-12


def test_direct_gather_kernels(oracle_lib, monkeypatch):
    """SMGPU_TILES=0: the non-tiled smoothing kernel leaves the boundary points to k_bnd_fix the same way"""
    from smoothmesh_amd.meshgen import hex_block
    monkeypatch.setenv("SMGPU_TILES", "0")
    m = tangential_jitter(hex_block(8, jitter=0.2, seed=2), 0.02, seed=1)
    init, target, surf = boundary_inputs(8, 3)
    for constraints in (False, True):
        o, e, prm, on = make_pair(m, oracle_lib, init, None, surf, constraints=constraints)
        _run_both(o, e, 6)


def test_find_line_matches_the_oracle(oracle_lib):
    """the bounding volume hierarchy returns what testing every triangle returns: random segments against a finely
    triangulated, warped surface, including segments that start on the surface and axis-parallel ones"""
    from smoothmesh_amd.meshgen import hex_block
    warp = lambda x: x + 0.03 * np.sin(5.0 * x[:, [1, 2, 0]])
    m = hex_block(4)
    init, _, _ = boundary_inputs(4, 2)
    from smoothmesh_amd.surfgen import box_surface
    surf = box_surface(12, warp=warp)
    o, e, prm, on = make_pair(m, oracle_lib, init, None, surf)
    rng = np.random.default_rng(0)
    a = rng.uniform(-0.3, 1.3, (400, 3))
    b = rng.uniform(-0.3, 1.3, (400, 3))
    b[:50] = a[:50] + np.eye(3)[rng.integers(0, 3, 50)] * rng.uniform(-1.5, 1.5, (50, 1))      # axis-parallel
    a[50:80] = surf[0][rng.integers(0, len(surf[0]), 30)]                                      # start on a surface vertex
    hit_g, p_g = e.debug_find_line(a, b)
    n_hit = 0
    for s in range(len(a)):
        hit_o, p_o = o.find_line(a[s], b[s])
        assert hit_o == hit_g[s]
        if hit_o:
            n_hit += 1
            assert np.array_equal(p_o, p_g[s])
    assert n_hit > 100


def test_persisted_classification_round_trip(oracle_lib):
    """the lists smgpu_get_boundary_classification returns, fed back, reproduce the run (SM.C:2039-2077)"""
    from smoothmesh_amd.meshgen import hex_block
    m = tangential_jitter(hex_block(7, jitter=0.2, seed=4), 0.02, seed=2)
    init, target, surf = boundary_inputs(7, 3)
    o, e, prm, on = make_pair(m, oracle_lib, init, None, surf)
    cio, fio = e.boundary_classification()
    e.iterate(5, 0.0)
    o2, e2, _, on2 = make_pair(m, oracle_lib, init, None, surf, cornerIO=cio, featureIO=fio)
    assert on2
    e2.iterate(5, 0.0)
    assert np.array_equal(e.get_points(), e2.get_points())


def test_not_available_with_a_halo(oracle_lib):
    from smoothmesh_amd import SmgpuError
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(4, jitter=0.1, seed=1)
    init, target, surf = boundary_inputs(4, 2)
    o, e, prm, on = make_pair(m, oracle_lib, init, None, surf)
    assert on
    # missing surface intersections are reported, not ignored (BPS.C:932-938): a target far away from the boundary
    far = (surf[0] * 1e-3 + 50.0, surf[1])
    o3, e3, _, on3 = make_pair(m, oracle_lib, init, None, far)
    with pytest.raises(SmgpuError, match="surface intersection"):
        e3.iterate(1, 0.0)
    with pytest.raises(RuntimeError):
        o3.iterate(1, 0.0)


def test_tiny_scale_cube_like_reference_testcase8(oracle_lib):
    """the reference's testcase8: a 3x3x3 block of 2e-8 m with its surface and twelve edges as targets, default options
    (`smoothMesh -centroidalIters 50`): every tolerance of the path is relative to the mesh size, nothing may depend on
    coordinates being O(1)"""
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    L = 2e-8
    m = hex_block(3, lengths=(L, L, L), jitter=0.25, seed=8)
    m.points[:] = np.array(m.points) - 0.5 * L
    lo, hi = (-0.5 * L,) * 3, (0.5 * L,) * 3
    init, surf = box_feature_edges(1, lo, hi), box_surface(1, lo, hi)           # one quad per side, as the reference's file
    o, e, prm, on = make_pair(m, oracle_lib, init, None, surf, constraints=True)
    assert on
    _check_setup(o, e)
    f = o.boundary_fields()
    assert f["isCornerPoint"].sum() == 8 and f["isFeatureEdgePoint"].sum() == 24
    res, frz = _run_both(o, e, 50)
    p = e.get_points()
    bnd = ~np.array(m.isInternalPoint, bool) if hasattr(m, "isInternalPoint") else (np.abs(np.abs(np.array(m.points)) - 0.5 * L).min(axis=1) == 0)
    assert np.all(np.abs(np.abs(p[bnd]) - 0.5 * L).min(axis=1) <= 1e-22)       # boundary points are still on the cube
    assert np.isfinite(res).all() and res[-1] < res[0]


def test_golden_fixture_boundary_hip(oracle_lib):
    """The HIP path against the committed golden vectors (tests/golden/make_golden_boundary.py; oracle-generated)."""
    import importlib.util
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_golden_boundary", os.path.join(here, "golden", "make_golden_boundary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(here, "golden", "hex6_boundary_seed5.npz"))
    m, init, target, surf = mod.case()
    o, e, prm, on = make_pair(m, oracle_lib, init, target, surf, constraints=True, blend=0.4)
    cio, fio = e.boundary_classification()
    assert np.array_equal(cio, g["isCornerPoint"]) and np.array_equal(fio, g["isFeatureEdgePoint"])
    frz_all, res_all = [], []
    for tag, iters in (("1", 1), ("5", 4), ("15", 10)):
        n, res, frz = e.iterate(iters, 0.0)
        frz_all.append(frz); res_all.append(res)
        assert rel_linf(e.get_points(), g["points" + tag]) <= 1e-13
    assert np.array_equal(np.concatenate(frz_all), g["nFrozen"])
    assert np.allclose(np.concatenate(res_all), g["residual"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("grid,constraints,layers", [((2, 1, 1), False, False), ((2, 2, 1), True, False), ((2, 2, 2), False, True),
                                                      ((1, 2, 2), True, True), ((3, 3, 3), False, False)])   # 3x3x3: one rank has no boundary
def test_decomposed_boundary_smoothing(oracle_lib, grid, constraints, layers):
    """-parallel: every engine holds one sub-domain; the set-up runs in steps with the reference's reductions and syncs
    between them, the per-iteration fields travel in the L records.  Bit-equal to the oracle's MultiDomain."""
    from smoothmesh_amd import BoundaryParams, LayerParams, patch_arrays
    from smoothmesh_amd.halo import LocalMultiSmoother
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    from test_oracle_boundary import _multi_boundary_case
    lpatches = ("xmin", "zmax") if layers else ()
    nloc = (3, 3, 4) if grid == (3, 3, 3) else (5, 4, 6)
    mo, orcs, subs, (off, dom, loc), hi = _multi_boundary_case(oracle_lib, grid, nloc, 0.25, constraints, blend=0.4, layerPatches=lpatches)
    prm = mo.params
    ms = LocalMultiSmoother(subs, device=0)
    ms.set_params(prm)
    if layers:
        assert ms.set_layers(LayerParams(layerPatches=lpatches), prm.minEdgeLength)
    bp = BoundaryParams(initEdges=box_feature_edges(8, hi=hi), targetSurfaces=box_surface(4, hi=hi), internalSmoothingBlendingFraction=0.4)
    infos = ms.set_boundary_smoothing(bp, prm.minEdgeLength)
    assert all(i["enabled"] for i in infos)
    for o, i in zip(orcs, infos):
        f = o.boundary_fields()
        assert i["nCornerPoints"] == f["isCornerPoint"].sum() and i["nFeatureEdgePoints"] == f["isFeatureEdgePoint"].sum()
        assert i["nSmoothingSurfacePoints"] == f["isSmoothingSurfacePoint"].sum()
    n_o, res_o, frz_o = mo.iterate(8, 0.0)
    n_g, res_g, frz_g = ms.iterate(8, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    for o, p in zip(orcs, ms.get_points()):
        assert rel_linf(p, o.points()) <= 1e-13


def test_distributed_smoother_with_boundary_smoothing_two_ranks():
    """One process per rank (torch.distributed.run), real engines, two ranks sharing this box's GPU through the gloo debug
    transport: DistributedSmoother with boundary point smoothing (and layers, constraints) equals the oracle's MultiDomain
    bit for bit (scripts/check_dist_boundary.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMOOTHMESH_SHARE_GPU="1", SMOOTHMESH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(root, "scripts", "check_dist_boundary.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ok ") == 4 and "BAD" not in r.stdout


@pytest.mark.parametrize("env", [{"SMGPU_HOST_WALK": "1"}, {"SMGPU_HOST_WALK": "0"}, {"SMGPU_WALK": "fix"}, {"SMGPU_FILTER": "0"},
                                 {"SMGPU_SIDE_STREAM": "0"}, {"SMGPU_STREAM_OPS": "0"}, {"SMGPU_XCD_MAP": "0"}, {"SMGPU_BND_IN_GEOM": "0"},
                                 {"SMGPU_BND_IN_GEOM": "0", "SMGPU_SIDE_STREAM": "0"}])
def test_boundary_smoothing_under_engine_knobs(oracle_lib, monkeypatch, env):
    """results never depend on the launch arrangement: walk replay place, filters, side streams, dependency mechanism"""
    from smoothmesh_amd.polymesh import cavity_mesh
    from smoothmesh_amd.surfgen import box_feature_edges, sphere_surface
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = cavity_mesh(10)
    o, e, prm, on = make_pair(m, oracle_lib, box_feature_edges(10), None, sphere_surface(levels=3), constraints=True,
                              smoothingPatches=("cavity",))
    assert on
    _run_both(o, e, 6)


@pytest.mark.parametrize("split", [False, True])
def test_baffle_with_boundary_point_smoothing_and_layers(oracle_lib, split):
    """the reference's testcase6 as a whole (run_serial:24: `-layerPatches '(walls "baffle.*")' -smoothingPatches '(".*")'` on a
    mesh with a baffle, createBaffles + splitBaffles): the wall inside the block has its own target surface (a plane through
    it) and its points start off that plane; with `split` the wall's interior points are
    coincident twins (a slit of zero width) that are classified, projected and layered independently"""
    from smoothmesh_amd.meshgen import add_baffle, baffle_in_plane, hex_block, split_baffles
    m = add_baffle(hex_block(10, 9, 8, jitter=0.2, seed=6), baffle_in_plane(hex_block(10, 9, 8), 0, 0.5, lambda c: (c[:, 1] < 0.7) & (c[:, 2] > 0.2)))
    if split:
        m = split_baffles(m)
    m = tangential_jitter(m, 0.02, seed=4)
    init, target, surf = boundary_inputs(10, 3)
    quad = np.array([[0.5, -0.1, 0.1], [0.5, 0.8, 0.1], [0.5, 0.8, 1.1], [0.5, -0.1, 1.1]])
    surf = (np.concatenate([surf[0], quad]), np.concatenate([surf[1], np.array([[0, 1, 2], [0, 2, 3]]) + len(surf[0])]))
    o, e, prm, on = make_pair(m, oracle_lib, init, target, surf, constraints=True, layerPatches=("ymax", '"baffle.*"'))
    assert on
    onwall = np.zeros(m.nPoints, bool)
    for p in m.patches:
        if p.name.startswith("baffle"):
            onwall[m.facePoints[m.faceOffsets[p.startFace]:m.faceOffsets[p.startFace + p.nFaces]]] = True
    assert np.abs(m.points[onwall, 0] - 0.5).max() > 1e-3          # off the plane at the start
    x0 = np.array(m.points).copy()
    n_o, res_o, frz_o = o.iterate(30, 0.0)
    n_g, res_g, frz_g = e.iterate(30, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= 1e-13
    assert np.abs(o.points()[onwall] - x0[onwall]).max() > 1e-3          # the wall's points take part (both forms; what they do is the oracle's)
