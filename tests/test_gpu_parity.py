"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded meshes.

Tolerances: coordinates 1e-10 relative L-inf is the north-star bar (BASELINE.json); the kernels keep
the reference's evaluation order, so we assert the much tighter 1e-13 and report exact-equality.
Integer outputs (nFrozenPoints series, frozen masks, addressing) must match exactly.
"""
import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu

COORD_TOL = 1e-13     # asserted; north-star tolerance is 1e-10
ANGLE_TOL = 0.0       # the kernels evaluate acos as a fixed sequence of IEEE operations (csrc/smacos.hpp) which the oracle repeats
                      # bit for bit when asked to (the fixture below); until round 4: 1e-12, glibc against the ROCm device library


@pytest.fixture(autouse=True)
def _oracle_evaluates_the_device_acos():
    """every angle field of the engine is compared with the oracle's BIT FOR BIT: the oracle then takes acos with the algorithm the
    kernels use (oracle_ffi.set_acos_variant("device")) instead of glibc's; tests/test_oracle_acos.py shows that the two variants
    lead to the same decisions and how far their angles are apart"""
    from oracle import oracle_ffi
    prev = oracle_ffi.set_acos_variant("device")
    yield
    oracle_ffi.set_acos_variant(prev)


def _mk(nx, ny, nz, jitter, seed):
    from smoothmesh_amd.meshgen import hex_block
    return hex_block(nx, ny, nz, jitter=jitter, seed=seed)


def _pair(mesh, oracle_lib, **over):
    from smoothmesh_amd import SmoothEngine, default_params
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    mn_o, mx_o = o.mesh_stats()
    mn_g, mx_g = e.mesh_stats()
    assert mn_o == mn_g and mx_o == mx_g          # sqrt + subtraction: bit-exact
    p = default_params(mn_o, **over)
    o.set_params(p)
    e.set_params(p)
    return o, e, p


CASES = [
    (8, 8, 8, 0.2, 1),
    (12, 9, 7, 0.3, 2),
    (5, 16, 6, 0.25, 3),
]


@pytest.mark.parametrize("nx,ny,nz,jit,seed", CASES)
def test_geometry_fields(oracle_lib, nx, ny, nz, jit, seed):
    mesh = _mk(nx, ny, nz, jit, seed)
    o, e, p = _pair(mesh, oracle_lib)
    o.phaseA(); o.phaseB()
    e.debug_propose()
    for name in ("faceCentres", "faceAreas", "cellCentres"):
        a, b = e.debug_field(name), o.field(name)
        assert rel_linf(a, b) <= 1e-15, name
    assert rel_linf(e.debug_field("newPoints"), o.field("newPoints")) <= COORD_TOL
    assert np.array_equal(e.debug_field("isFrozenPoint"), o.field("frozenAfterFaceAngle"))
    for name in ("edgeMinAngle", "edgeMaxAngle", "pointMinAngle", "pointMaxAngle"):
        assert np.max(np.abs(e.debug_field(name) - o.field(name))) <= ANGLE_TOL, name


@pytest.mark.parametrize("nx,ny,nz,jit,seed", CASES)
@pytest.mark.parametrize("constraints", [False, True])
def test_iterations_match_oracle(oracle_lib, nx, ny, nz, jit, seed, constraints):
    mesh = _mk(nx, ny, nz, jit, seed)
    o, e, p = _pair(mesh, oracle_lib, edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    n_o, res_o, frz_o = o.iterate(20, 0.0)
    n_g, res_g, frz_g = e.iterate(20, 0.0)
    assert n_o == n_g == 20
    assert np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


def test_relTol_stops_like_reference(oracle_lib):
    mesh = _mk(8, 8, 8, 0.2, 5)
    o, e, p = _pair(mesh, oracle_lib, edgeAngleConstraint=False, faceAngleConstraint=False)
    n_o, res_o, _ = o.iterate(200, 0.02)
    n_g, res_g, _ = e.iterate(200, 0.02)
    assert n_o == n_g and n_o < 200
    assert res_g[-1] < 0.02
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


@pytest.mark.parametrize("swap", ["1", "0"])
def test_restore_step_forms_agree(oracle_lib, monkeypatch, swap):
    """constraints on, single rank: the proposal array becomes the next coordinates by a pointer swap and k_apply_swap only
    restores the points that do not move (SM.C:2384-2399; SMGPU_APPLY_SWAP=0: k_apply copies everything into a third array).
    Odd and even iteration counts per call, a relTol stop in the middle of a call, coordinates read and written between calls --
    the residual series, the frozen counts and the coordinates are the oracle's in both forms."""
    monkeypatch.setenv("SMGPU_APPLY_SWAP", swap)
    mesh = _mk(9, 8, 7, 0.47, 21)     # (the constraints freeze ~20 interior points per iteration)
    o, e, p = _pair(mesh, oracle_lib, edgeAngleConstraint=True, faceAngleConstraint=True)
    for k in (3, 1, 4):
        n_o, res_o, frz_o = o.iterate(k, 0.0)
        n_g, res_g, frz_g = e.iterate(k, 0.0)
        assert n_o == n_g == k
        assert np.array_equal(frz_o, frz_g) and np.array_equal(res_o, res_g)
        assert np.array_equal(e.get_points(), o.points())
        assert np.array_equal(e.debug_field("points").reshape(-1, 3), o.points())
    # a stop inside the call: the launches behind it must leave both arrays alone
    n_o, res_o, frz_o = o.iterate(60, 0.2)
    n_g, res_g, frz_g = e.iterate(60, 0.2)
    assert n_o == n_g and 1 <= n_o < 60, n_o
    assert np.array_equal(res_o, res_g) and np.array_equal(frz_o, frz_g)
    assert np.array_equal(e.get_points(), o.points())
    # new coordinates from the host, then on
    pts = o.points().copy()
    pts[mesh.nPoints // 2] += 1e-3
    o.set_points(pts); e.set_points(pts)
    n_o, res_o, frz_o = o.iterate(5, 0.0)
    n_g, res_g, frz_g = e.iterate(5, 0.0)
    assert np.array_equal(res_o, res_g) and np.array_equal(frz_o, frz_g)
    assert np.array_equal(e.get_points(), o.points())


def test_uniform_block_is_fixed_point(oracle_lib):
    # h = 1/8 is exact in binary, so every cell centre / centroid is exact: residual exactly 0
    mesh = _mk(8, 8, 8, 0.0, 0)
    o, e, p = _pair(mesh, oracle_lib)
    n, res, frz = e.iterate(10, 0.02)
    assert n == 1 and res[0] == 0.0          # stops after 1 iteration (SM.C:2401)
    assert frz[0] == mesh.nPoints - 7 ** 3   # every boundary point counts as frozen (SM.C:2387-2391)
    assert np.array_equal(e.get_points(), mesh.points)
    # h = 1/6 is not exact: residual is rounding noise, identical to the oracle's
    mesh = _mk(6, 6, 6, 0.0, 0)
    o, e, p = _pair(mesh, oracle_lib)
    n_o, res_o, frz_o = o.iterate(10, 0.02)
    n_g, res_g, frz_g = e.iterate(10, 0.02)
    assert n_o == n_g == 1 and res_o[0] == res_g[0] and res_g[0] < 1e-12
    assert np.array_equal(frz_o, frz_g)


@pytest.mark.parametrize("walk", ["auto", "wave", "host", "fix"])
@pytest.mark.parametrize("jit,seed", [(0.45, 7), (0.48, 11)])
def test_bad_mesh_face_angle_walk(oracle_lib, monkeypatch, jit, seed, walk):
    """Heavy jitter pushes face angles outside [35, 160] degrees: the ordered freeze walk
    (SM.C:1347-1434) is exercised, self- and neighbour-freezes included -- in every place the replay can run
    (one wave over the flag array, the host, the causal fixed point in one persistent launch)."""
    if walk != "auto":
        monkeypatch.setenv("SMGPU_WALK", walk)
    mesh = _mk(8, 7, 6, jit, seed)
    o, e, p = _pair(mesh, oracle_lib)
    o.phaseA(); o.phaseB()
    e.debug_propose()
    fo = o.field("frozenAfterFaceAngle")
    assert fo.sum() > o.field("frozenAfterEdgeAngle").sum()   # the walk froze something
    assert np.array_equal(e.debug_field("isFrozenPoint"), fo)
    n_o, res_o, frz_o = o.iterate(10, 0.0)
    n_g, res_g, frz_g = e.iterate(10, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


def test_one_wave_walk_behind_many_workgroups(oracle_lib, monkeypatch):
    """The one-wave replay forced on a mesh of several hundred workgroups, most of them with active points (the form the
    engine picks by itself only when few points are active)."""
    monkeypatch.setenv("SMGPU_WALK", "wave")
    mesh = _mk(44, 40, 36, 0.46, 5)
    o, e, p = _pair(mesh, oracle_lib)
    n_o, res_o, frz_o = o.iterate(4, 0.0)
    n_g, res_g, frz_g = e.iterate(4, 0.0)
    assert frz_o[0] > 1000
    assert np.array_equal(frz_o, frz_g)
    assert np.array_equal(e.get_points(), o.points())


@pytest.mark.parametrize("knobs", [{}, {"SMGPU_WALK_PACK": "0"}, {"SMGPU_WALK_WARM": "0"}, {"SMGPU_WALK_LOCAL": "0"}, {"SMGPU_WALK_SWEEPS": "1"}, {"share": 4},
                                   {"SMGPU_WALK_CACHE": "0"}, {"SMGPU_WALK_CACHE_CAP": "40"}])
@pytest.mark.parametrize("dims,jit,seed", [((14, 12, 10), 0.47, 3), ((20, 6, 5), 0.49, 8)])
def test_fixed_point_walk_on_large_components(oracle_lib, monkeypatch, dims, jit, seed, knobs):
    """a badly distorted block: the interaction graph has components of hundreds of points with long re-visit chains;
    the fixed-point device replay must still reproduce the reference's order -- from the previous iteration's set or from the
    empty one, with the sweeps in LDS or in global memory, one or several sweeps between two grid barriers, with the full
    persistent launch or a quarter of it (smgpu_set_device_share); with the stars' static records (the default), without them
    (every star staged from the addressing every iteration), and with a pool of 40 records (most points staged, a few from their
    records, in the same launch group).  Coordinates are written from the host in the middle (the
    warm start then begins from a set that belongs to other coordinates)."""
    monkeypatch.setenv("SMGPU_WALK", "fix")
    for k, v in knobs.items():
        if k.startswith("SMGPU_"):
            monkeypatch.setenv(k, v)
    mesh = _mk(*dims, jit, seed)
    o, e, p = _pair(mesh, oracle_lib)
    if "share" in knobs:
        e.set_device_share(knobs["share"])
    n_o, res_o, frz_o = o.iterate(6, 0.0)
    n_g, res_g, frz_g = e.iterate(6, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL
    o.set_points(mesh.points); e.set_points(mesh.points)
    n_o, res_o, frz_o = o.iterate(3, 0.0)
    n_g, res_g, frz_g = e.iterate(3, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


def test_walk_replay_form_follows_the_mesh_mid_run(oracle_lib, monkeypatch):
    """The replay form of the walk is re-decided while the run goes (smgpu.hip:updateWalkMode), from the count of points outside
    the good range that the GPU publishes at the end of every iteration.  (a) A distorted block heals: 3 040 such points at the
    start, < 500 after six iterations -- with the threshold at 1 000 the run starts on the fixed-point replay and must come
    down to the one-wave replay.  (b) A good block is overwritten with the distorted coordinates mid-run (smgpu_set_points, no
    new parameters): the one-wave replay must hand over to the fixed-point replay.  Results equal the oracle's either way."""
    monkeypatch.setenv("SMGPU_HOST_WALK_THRESHOLD", "1000")
    bad = _mk(24, 24, 24, 0.45, 5)
    o, e, p = _pair(bad, oracle_lib)
    n_o, res_o, frz_o = o.iterate(40, 0.0)
    n_g, res_g, frz_g = e.iterate(40, 0.0)
    mode, switches = e.debug_walk_mode()
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL
    assert mode == 0 and switches >= 1, (mode, switches)
    e.close()
    # (b): with this block's own step lengths the distorted coordinates heal within a few iterations (3 040, 660, 223, 155, ...
    # points; 86 after thirty iterations), and the host sees the counts with a lag of up to 16 iterations: threshold 60, so that
    # the count stays above it for the whole call
    monkeypatch.setenv("SMGPU_HOST_WALK_THRESHOLD", "60")
    good = _mk(24, 24, 24, 0.2, 5)
    o, e, p = _pair(good, oracle_lib)
    o.iterate(3, 0.0); e.iterate(3, 0.0)
    assert e.debug_walk_mode() == (0, 0)
    o.set_points(bad.points); e.set_points(bad.points)
    n_o, res_o, frz_o = o.iterate(24, 0.0)
    n_g, res_g, frz_g = e.iterate(24, 0.0)
    mode, switches = e.debug_walk_mode()
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL
    assert switches >= 1, (mode, switches)      # went up to the fixed-point replay (and possibly down again)


def test_totalMinFreeze_and_explicit_lengths(oracle_lib):
    mesh = _mk(7, 7, 7, 0.3, 13)
    o, e, p = _pair(mesh, oracle_lib, totalMinFreeze=True, minEdgeLength=0.11, maxStepLength=0.004, minAngle=50.0, maxAngle=130.0)
    n_o, res_o, frz_o = o.iterate(8, 0.0)
    n_g, res_g, frz_g = e.iterate(8, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert frz_o.max() > mesh.nPoints - 6 ** 3      # some interior points got frozen
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


def test_run_to_run_bitwise_repeatable(oracle_lib):
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = _mk(10, 10, 10, 0.3, 21)
    outs = []
    for _ in range(2):
        e = SmoothEngine(mesh)
        e.set_params(default_params(e.mesh_stats()[0]))
        e.iterate(15, 0.0)
        outs.append(e.get_points())
        e.close()
    assert np.array_equal(outs[0], outs[1])


def test_large_mesh_properties():
    """Full-size config (100^3, BASELINE configs[1]): size-independent properties -- boundary points never
    move, the residual series is finite/non-negative, every interior step is bounded by maxStepLength."""
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = _mk(100, 100, 100, 0.2, 12345)
    e = SmoothEngine(mesh)
    p = default_params(e.mesh_stats()[0], edgeAngleConstraint=False, faceAngleConstraint=False)
    e.set_params(p)
    before = mesh.points.copy()
    n, res, frz = e.iterate(10, 0.0)
    after = e.get_points()
    assert n == 10 and np.all(np.isfinite(res)) and np.all(res >= 0) and np.all(res <= 1.0 + 1e-12)
    internal = mesh.find_internal_points().astype(bool)
    assert np.array_equal(after[~internal], before[~internal])
    assert np.max(np.linalg.norm(after - before, axis=1)) <= 10 * p.maxStepLength * (1 + 1e-12)
    assert np.all(frz >= (~internal).sum())


@pytest.mark.parametrize("jitter,minAngle", [(0.2, 35.0), (0.42, 60.0)])
def test_large_mesh_with_constraints_filters_are_conservative(monkeypatch, jitter, minAngle):
    """Full-size config (100^3 with the edge- and face-angle constraints on, BASELINE configs[2]; the second case with heavier
    jitter and a larger minAngle so that the evaluators do freeze thousands of points): the f32 filters, the list-based exact
    pass and the walk mode chosen at this size decide exactly what the unfiltered exact kernels decide -- identical
    nFrozenPoints / residual series and coordinates; boundary points never move; steps stay within maxStepLength."""
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = _mk(100, 100, 100, jitter, 12345)
    internal = mesh.find_internal_points().astype(bool)
    outs = []
    for filt in ("1", "0"):
        monkeypatch.setenv("SMGPU_FILTER", filt)
        e = SmoothEngine(mesh)
        p = default_params(e.mesh_stats()[0], minAngle=minAngle)
        e.set_params(p)
        n, res, frz = e.iterate(12, 0.0)
        outs.append((res, frz, e.get_points()))
        e.close()
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][2], outs[1][2])
    res, frz, after = outs[0]
    assert np.all(np.isfinite(res)) and np.all(res >= 0) and np.all(res <= 1.0 + 1e-12)
    assert np.array_equal(after[~internal], mesh.points[~internal])
    assert np.max(np.linalg.norm(after - mesh.points, axis=1)) <= 12 * p.maxStepLength * (1 + 1e-12)
    assert np.all(frz >= (~internal).sum())
    if jitter > 0.4:
        assert frz[0] > (~internal).sum() + 1000       # the constraints are busy


def test_golden_fixture_hip():
    """The HIP path against the committed golden vectors (tests/golden/make_golden.py; oracle-generated)."""
    import os
    from smoothmesh_amd import SmoothEngine, default_params
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hex6_jitter03_seed7.npz"))
    mesh = _mk(6, 6, 6, 0.3, 7)
    assert np.array_equal(mesh.points, g["points0"])
    e = SmoothEngine(mesh)
    e.set_params(default_params(e.mesh_stats()[0]))
    frz_all, res_all = [], []
    for tag, iters in (("1", 1), ("5", 4), ("20", 15)):
        n, res, frz = e.iterate(iters, 0.0)
        frz_all.append(frz); res_all.append(res)
        assert rel_linf(e.get_points(), g["points" + tag]) <= COORD_TOL
    assert np.array_equal(np.concatenate(frz_all), g["nFrozen"])
    assert np.allclose(np.concatenate(res_all), g["residual"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("jit,minAngle", [(0.3, 35.0), (0.45, 60.0), (0.48, 75.0)])
def test_edge_angle_forms_agree(oracle_lib, monkeypatch, jit, minAngle):
    """The wave-cooperative edge-angle kernel (acos of the extreme cosine) must freeze exactly the points the
    per-angle form (5 acos per corner, SM.C:862-880) and the oracle freeze."""
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = _mk(9, 8, 7, jit, 17)
    o = oracle_lib.Oracle(mesh)
    p = default_params(o.mesh_stats()[0], minAngle=minAngle, faceAngleConstraint=False)
    o.set_params(p)
    o.phaseA(); o.phaseB()
    ref = o.field("frozenAfterEdgeAngle")
    assert ref.sum() > o.field("frozenAfterEdgeLen").sum() or minAngle < 40     # the evaluator froze something
    masks = []
    for mode in ("coop", "faithful"):
        monkeypatch.setenv("SMGPU_EDGE_ANGLE", mode)
        e = SmoothEngine(mesh)
        e.set_params(p)
        e.debug_propose()
        masks.append(e.debug_field("isFrozenPoint"))
        e.close()
    assert np.array_equal(masks[0], masks[1])
    assert np.array_equal(masks[0], ref)


def test_walk_replay_places_agree_at_scale(monkeypatch):
    """cavity100c (1 M cells, ~25 k acting points in ~10 k components per iteration): the fixed-point device replay, the
    host replay and nothing else changed -- identical nFrozenPoints series and coordinates after 12 iterations, and the
    frozen set only ever contains points the walk or the other evaluators may freeze (never a non-moving one wrongly freed)"""
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.polymesh import cavity_mesh
    mesh = cavity_mesh(100, jitter=0.2, seed=12345)
    outs = {}
    for walk in ("fix", "host"):
        monkeypatch.setenv("SMGPU_WALK", walk)
        e = SmoothEngine(mesh)
        e.set_params(default_params(e.mesh_stats()[0]))
        n, res, frz = e.iterate(12, 0.0)
        outs[walk] = (res, frz, e.get_points())
        e.close()
    assert np.array_equal(outs["fix"][1], outs["host"][1])
    assert np.array_equal(outs["fix"][0], outs["host"][0])
    assert np.array_equal(outs["fix"][2], outs["host"][2])
    assert outs["fix"][1][-1] > outs["fix"][1][0] > (~mesh.find_internal_points().astype(bool)).sum()   # the walk freezes more and more


@pytest.mark.parametrize("walk", ["auto", "fix"])
@pytest.mark.parametrize("constraints", [False, True])
def test_polyhedral_cavity_mesh(oracle_lib, monkeypatch, constraints, walk):
    """Castellated octree mesh (polyhedral cells with split faces and hanging edge nodes, valence 3..6,
    coplanar face pairs => face angles of 180 degrees => the ordered freeze walk is busy every iteration)."""
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.polymesh import cavity_mesh
    if walk != "auto":
        if not constraints:
            pytest.skip("no walk without the constraints")
        monkeypatch.setenv("SMGPU_WALK", walk)
    mesh = cavity_mesh(12, jitter=0.2, seed=3)
    assert np.bincount(np.diff(mesh.faceOffsets))[5:].sum() > 0          # genuinely polygonal faces
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    p = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(p); e.set_params(p)
    if constraints:
        o.phaseA(); o.phaseB()
        e.debug_propose()
        assert o.field("frozenAfterFaceAngle").sum() > o.field("frozenAfterEdgeAngle").sum()
        assert np.array_equal(e.debug_field("isFrozenPoint"), o.field("frozenAfterFaceAngle"))
        for name in ("edgeMinAngle", "edgeMaxAngle", "pointMinAngle", "pointMaxAngle"):
            assert np.max(np.abs(e.debug_field(name) - o.field(name))) <= ANGLE_TOL, name
    n_o, res_o, frz_o = o.iterate(12, 0.0)
    n_g, res_g, frz_g = e.iterate(12, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= COORD_TOL


@pytest.mark.parametrize("kind,arg,jit", [("hex", (9, 8, 7), 0.3), ("hex", (8, 7, 6), 0.47), ("cavity", 10, 0.2)])
def test_filtered_path_makes_the_same_decisions(oracle_lib, monkeypatch, kind, arg, jit):
    """The f32 filters (kernels_filter.hpp) only skip elements that are certainly far from the thresholds: the
    active set and the frozen set after the three evaluators must equal the unfiltered run's and the oracle's."""
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.polymesh import cavity_mesh
    mesh = _mk(*arg, jit, 23) if kind == "hex" else cavity_mesh(arg, jitter=jit, seed=23)
    o = oracle_lib.Oracle(mesh)
    p = default_params(o.mesh_stats()[0], minAngle=45.0, maxAngle=150.0)
    o.set_params(p)
    o.phaseA(); o.phaseB()
    small, large = np.pi * 45 / 180, np.pi * 150 / 180
    act_ref = ~((o.field("pointMinAngle") > small) & (o.field("pointMaxAngle") < large))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SMGPU_DEBUG_FILTERED", mode)
        e = SmoothEngine(mesh)
        e.set_params(p)
        e.debug_propose()
        out[mode] = (e.debug_field("isFrozenPoint"), e.debug_field("faActive"))
        e.close()
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])
    assert np.array_equal(out["1"][0], o.field("frozenAfterFaceAngle"))
    assert np.array_equal(out["1"][1].astype(bool), act_ref)


@pytest.mark.parametrize("A,frac", [(0.4375, 0.0), (0.703125, 0.5), (0.8203125, 0.75), (1.0, 1.0)])
def test_aspect_ratio_blend_known_answers_on_the_gpu(A, frac):
    """The analytic answers of tests/test_oracle_known_answers.py::test_aspect_ratio_blend_known_answers asked of the ENGINE itself
    (no oracle in between): box cells on binary fractions, the interior point of a 2x2x2 block ends at
    z = (1 - f) (s2 - s1) / 4 + f (s2 - s1) / 2 with f = clamp((A / s2 - 1.5) / 1.5) (SM.C:489-543, 548-591), exactly."""
    from test_oracle_known_answers import _lattice_2x2x2
    from smoothmesh_amd import SmoothEngine, SmoothParams
    s1, s2 = 0.25, 0.3125
    m = _lattice_2x2x2((-A, 0.0, A), (-A, 0.0, A), (-s1, 0.0, s2))
    e = SmoothEngine(m)
    e.set_params(SmoothParams(maxStepLength=10.0, minEdgeLength=1e-6, relStepFrac=1.0, edgeAngleConstraint=False, faceAngleConstraint=False))
    n, res, frz = e.iterate(1, 0.0)
    z = (1.0 - frac) * (s2 - s1) / 4 + frac * (s2 - s1) / 2
    got = e.get_points()
    assert np.allclose(got[13], [0.0, 0.0, z], atol=1e-16) and frz[0] == 26 and abs(res[0] - z / 10.0) <= 1e-17
    assert np.array_equal(np.delete(got, 13, axis=0), np.delete(np.array(m.points), 13, axis=0))     # boundary points stay
