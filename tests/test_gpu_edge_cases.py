"""Edge cases of the HIP path: tiny and ragged meshes, high valence (wide ELL rows, pair-table fallback),
polygon faces with many vertices, the direct-gather fallback kernels, error reporting."""
import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu


def _fan_mesh(nSpokes, nLayers=3, jitter=0.05, seed=1):
    """prisms around an axis: the axis points have valence nSpokes + 2"""
    from smoothmesh_amd.meshgen import extrude_surface
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, nSpokes, endpoint=False)
    ring1 = np.stack([np.cos(ang), np.zeros(nSpokes), np.sin(ang)], axis=1)
    ring2 = 2.0 * np.stack([np.cos(ang + 0.1), np.zeros(nSpokes), np.sin(ang + 0.1)], axis=1)
    verts = np.concatenate([[[0.0, 0.0, 0.0]], ring1, ring2])
    verts[1:1 + nSpokes] += jitter * rng.standard_normal((nSpokes, 3)) * [1, 0, 1]
    faces = []
    for i in range(nSpokes):
        j = (i + 1) % nSpokes
        faces.append([0, 1 + i, 1 + j])                                   # inner triangles
        faces.append([1 + i, 1 + nSpokes + i, 1 + nSpokes + j, 1 + j])    # outer quads
    return extrude_surface(verts, faces, nLayers=nLayers, thickness=1.0, direction=(0, 1, 0))


def _compare(mesh, oracle_lib, iters=6, **over):
    from smoothmesh_amd import SmoothEngine, default_params
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    p = default_params(o.mesh_stats()[0], **over)
    o.set_params(p); e.set_params(p)
    n_o, res_o, frz_o = o.iterate(iters, 0.0)
    n_g, res_g, frz_g = e.iterate(iters, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= 1e-13
    return e


@pytest.mark.parametrize("dims", [(1, 1, 1), (2, 1, 1), (3, 2, 1), (2, 2, 2)])
def test_tiny_blocks(oracle_lib, dims):
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(*dims, jitter=0.2, seed=3)
    _compare(m, oracle_lib)
    _compare(m, oracle_lib, edgeAngleConstraint=False, faceAngleConstraint=False)


@pytest.mark.parametrize("nSpokes", [5, 12, 20])
def test_high_valence_fan(oracle_lib, nSpokes):
    """valence nSpokes+2 on the axis: ELL rows wider than 8 entries; > 16 switches hasCommonCell to the
    pointCells-intersection form (no pair table)"""
    m = _fan_mesh(nSpokes)
    _compare(m, oracle_lib, iters=8)
    _compare(m, oracle_lib, iters=8, edgeAngleConstraint=False, faceAngleConstraint=False, minEdgeLength=0.3, totalMinFreeze=True)


def test_zero_iterations_and_param_errors(oracle_lib):
    from smoothmesh_amd import SmgpuError, SmoothEngine, SmoothParams, default_params
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(3, jitter=0.2)
    e = SmoothEngine(m)
    with pytest.raises(SmgpuError, match="set_params"):
        e.iterate(1, 0.0)
    with pytest.raises(SmgpuError, match="maxStepLength"):
        e.set_params(SmoothParams(maxStepLength=0.0, minEdgeLength=0.1))
    e.set_params(default_params(e.mesh_stats()[0]))
    n, res, frz = e.iterate(0, 0.0)
    assert n == 0 and len(res) == 0
    assert np.array_equal(e.get_points(), m.points)
    pts = m.points + 0.001
    e.set_points(pts)
    assert np.array_equal(e.get_points(), pts)


def test_direct_gather_fallback_kernels(oracle_lib, monkeypatch):
    """SMGPU_TILES=0: the one-thread-per-element kernels (used when a mesh does not fit the LDS tile tables)"""
    from smoothmesh_amd.polymesh import cavity_mesh
    monkeypatch.setenv("SMGPU_TILES", "0")
    m = cavity_mesh(8, jitter=0.2, seed=5)
    _compare(m, oracle_lib, iters=5)
    _compare(m, oracle_lib, iters=5, edgeAngleConstraint=False, faceAngleConstraint=False)


def test_natural_tile_order(oracle_lib, monkeypatch):
    from smoothmesh_amd.meshgen import hex_block
    monkeypatch.setenv("SMGPU_TILE_MORTON", "0")
    _compare(hex_block(9, 7, 5, jitter=0.3, seed=2), oracle_lib)


@pytest.mark.parametrize("env", [{"SMGPU_XCD_MAP": "0"}, {"SMGPU_WALK_STAR": "0"}, {"SMGPU_WALK_PACK": "0"}, {"SMGPU_TILE_MORTON": "0"}, {"SMGPU_FA_SIDE_EXACT": "0"},
                                 {"SMGPU_SIDE_STREAM": "0"}, {"SMGPU_WALK_CACHE": "0"}, {"SMGPU_WALK_CACHE_CAP": "100"}])
def test_launch_variants_give_the_same_result(oracle_lib, monkeypatch, env):
    """launch arrangements (round-robin tile launch, natural tile order, gather-form walk predicates, exact face-angle pass on
    the main stream, no side streams): tuning knobs, same bits; meshes with quadrilateral-only and mixed tiles"""
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import cavity_mesh
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _compare(hex_block(24, 20, 18, jitter=0.25, seed=8), oracle_lib, iters=4, edgeAngleConstraint=False, faceAngleConstraint=False)
    _compare(cavity_mesh(16, jitter=0.2, seed=3), oracle_lib, iters=3)


@pytest.mark.parametrize("case", ["coincident", "collapsed_cell", "inverted"])
@pytest.mark.parametrize("constraints", [False, True])
def test_degenerate_geometry_matches_the_oracle(oracle_lib, case, constraints):
    """zero-length edges (NaN cosines -> clamp rule SM.C:781), a zero-volume cell (|V| <= VSMALL -> face-centre average),
    inverted cells: the same special-case branches on both sides, the same bits out"""
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(6, 5, 4, jitter=0.2, seed=3)
    P, n1 = m.points, 7
    p = (n1 * 6) * 2 + n1 * 3 + 3                  # an interior point
    if case == "coincident":
        P[p] = P[p + 1]
    elif case == "collapsed_cell":
        for q in (p, p + 1, p + n1, p + n1 + 1):
            P[q] = P[q + n1 * 6]
    else:
        P[p] = P[p] + (P[p + 1] - P[p]) * 1.7
    o = oracle_lib.Oracle(m)
    e = SmoothEngine(m)
    mn = o.mesh_stats()[0]
    assert mn == e.mesh_stats()[0]
    prm = default_params(max(mn, 1e-3), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(prm); e.set_params(prm)
    n_o, res_o, frz_o = o.iterate(4, 0.0)
    n_g, res_g, frz_g = e.iterate(4, 0.0)
    assert np.array_equal(frz_o, frz_g)
    a, b = e.get_points(), o.points()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~np.isnan(b)], b[~np.isnan(b)])


def test_counters_and_sizes(oracle_lib):
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(10, jitter=0.2)
    e = SmoothEngine(m)
    sz = e.sizes()
    assert sz["nPoints"] == 1331 and sz["nCells"] == 1000 and sz["nEdges"] == 3 * 10 * 11 * 11
    assert sz["nnzPointCells"] == 8 * 1000 and sz["nnzFacePoints"] == 4 * sz["nFaces"]
    e.set_params(default_params(e.mesh_stats()[0]))
    e.enable_timing(True)
    e.iterate(5, 0.0)
    c = {k["name"]: k for k in e.counters()}
    assert c["k_geom_tile"]["launches"] == 5 and c["k_geom_tile"]["ms"] > 0
    assert c["k_smooth<proposal>"]["algoBytesPerLaunch"] > 100 * 1331


@pytest.mark.parametrize("mesh_kind", ["hex", "cavity", "nonconvex", "degenerate"])
@pytest.mark.parametrize("tiles", ["1", "0"])
def test_openfoam_org_geometry_variant(oracle_lib, monkeypatch, mesh_kind, tiles):
    """OpenFOAM.org 12's makeFaceCentresAndAreas / makeCellCentresAndVols (fan triangles weighted by the projected area,
    pyramids clamped at vSmall) as a run-time switch: tiled and direct kernels against the oracle's restatement; geometry
    fields to rounding, coordinates bit for bit; and the results do differ from the OpenFOAM.com default"""
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import cavity_mesh
    monkeypatch.setenv("SMGPU_TILES", tiles)
    if mesh_kind == "hex":
        mesh = hex_block(11, 9, 8, jitter=0.3, seed=4)
    elif mesh_kind == "cavity":
        mesh = cavity_mesh(12, jitter=0.25, seed=6)
    elif mesh_kind == "nonconvex":
        mesh = cavity_mesh(8, jitter=0.45, seed=2)          # heavy jitter on 5..8-vertex faces: non-convex, strongly warped polygons
    else:
        mesh = hex_block(4, 4, 3, jitter=0.2, seed=1)
        mesh.points[mesh.facePoints[mesh.faceOffsets[40]:mesh.faceOffsets[41]]] = mesh.points[mesh.facePoints[mesh.faceOffsets[40]]]   # a face collapsed to a point
    o = oracle_lib.Oracle(mesh)
    o.set_foam_variant("org")
    e = SmoothEngine(mesh)
    e.set_foam_variant("org")
    p = default_params(o.mesh_stats()[0], minEdgeLength=1e-3 if mesh_kind == "degenerate" else 0.5 * o.mesh_stats()[0])
    o.set_params(p); e.set_params(p)
    o.phaseA(); o.phaseB()
    e.debug_propose()
    for name in ("faceCentres", "faceAreas", "cellCentres"):
        a, b = e.debug_field(name), o.field(name)
        ok = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), ok)
        assert np.max(np.abs(a[ok] - b[ok])) <= 1e-15 * max(1.0, np.max(np.abs(b[ok]))), name
    n_o, res_o, frz_o = o.iterate(5, 0.0)
    n_g, res_g, frz_g = e.iterate(5, 0.0)
    assert np.array_equal(frz_o, frz_g)
    ok = np.isfinite(o.points())
    assert np.array_equal(e.get_points()[ok], o.points()[ok])
    if mesh_kind in ("hex", "cavity"):
        e2 = SmoothEngine(mesh)                             # default variant: OpenFOAM.com
        e2.set_params(p)
        e2.iterate(5, 0.0)
        assert not np.array_equal(e2.get_points(), e.get_points())


@pytest.mark.parametrize("angles", [(0.0, 180.0), (0.0, 90.0), (89.0, 91.0), (35.0, 179.9), (-5.0, 200.0), (60.0, 120.0), (120.0, 100.0),
                                    (179.0, 180.0), (35.0, 0.0)])
@pytest.mark.parametrize("kind", ["hex", "cavity"])
def test_face_angle_filter_at_extreme_thresholds(oracle_lib, kind, angles):
    """The f32 face-angle filter decides on the cosine of an angle sum against host-prepared thresholds (kernels_filter.hpp):
    thresholds at and beyond the ends of [0, 180] degrees, a two-degree band around the block's own right angles (every edge
    borderline), a band that excludes them, an empty and an inverted range -- the frozen sets must stay those of the oracle."""
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import cavity_mesh
    m = hex_block(9, 8, 7, jitter=0.3, seed=4) if kind == "hex" else cavity_mesh(10, jitter=0.25, seed=4)
    _compare(m, oracle_lib, iters=4, minAngle=angles[0], maxAngle=angles[1])


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("variant", ["defaults", "busy", "layers", "no-constraints", "walk-fix"])
def test_baffle_inside_the_block(oracle_lib, monkeypatch, variant, split):
    """A zero-thickness wall inside the mesh (createBaffles; the reference's testcase6): pairs of boundary faces on the SAME points
    (or, split, on coincident twins), edges whose face ring is cut open by the wall, boundary points with cells on both sides -- constraints on and busy, layers grown
    from both sides of the baffle (`-layerPatches '("baffle.*")'`, testcase6/run_serial:24), the fixed-point walk."""
    from smoothmesh_amd import LayerParams, SmoothEngine, default_params, patch_arrays
    from smoothmesh_amd.meshgen import add_baffle, baffle_in_plane, hex_block, split_baffles
    jit = 0.45 if variant in ("busy", "walk-fix") else 0.25
    m = add_baffle(hex_block(10, 9, 8, jitter=jit, seed=6), baffle_in_plane(hex_block(10, 9, 8), 0, 0.5, lambda c: (c[:, 1] < 0.7) & (c[:, 2] > 0.2)))
    assert [p.name for p in m.patches][-2:] == ["baffle_master", "baffle_slave"]
    if split:      # splitBaffles (testcase6/run_serial:17-18): the wall's interior points twinned, a slit of zero width
        nP = m.nPoints
        m = split_baffles(m)
        assert m.nPoints == nP + 36
    if variant == "walk-fix":
        monkeypatch.setenv("SMGPU_WALK", "fix")
    over = {"busy": dict(minAngle=50.0), "walk-fix": dict(minAngle=50.0), "no-constraints": dict(edgeAngleConstraint=False, faceAngleConstraint=False)}.get(variant, {})
    o = oracle_lib.Oracle(m)
    e = SmoothEngine(m)
    prm = default_params(o.mesh_stats()[0], **over)
    o.set_params(prm); e.set_params(prm)
    if variant == "layers":
        L = LayerParams(layerPatches=('"baffle.*"',), layerExpansionRatio=1.2, maxLayers=3)
        st, sz, kd, sel = patch_arrays(m, L.layerPatches)
        assert sel.sum() == 2
        assert o.setup_layers(st, sz, kd, sel, L.layerMaxBlendingFraction, prm.minEdgeLength, 1.2, 1, 3) == e.set_layers(L, prm.minEdgeLength)
    n_o, res_o, frz_o = o.iterate(8, 0.0)
    n_g, res_g, frz_g = e.iterate(8, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    if variant in ("busy", "walk-fix"):
        assert frz_o[0] > int((1 - m.find_internal_points()).sum()) + 20      # the constraints froze interior points
    assert np.array_equal(e.get_points(), o.points())
