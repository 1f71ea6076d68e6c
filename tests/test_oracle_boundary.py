"""Known answers for the oracle's boundary point smoothing (BPS.C = src/boundaryPointSmoothing.C, OBB.C:573-631, set-up
SM.C:2080-2253, per iteration SM.C:2266 + 2307-2357) on cases where the reference's result can be written down."""
import numpy as np
import pytest

from bnd_cases import boundary_inputs, make_pair, scale_about_centre, tangential_jitter


def _ijk(n):
    p = np.arange((n + 1) ** 3)
    return p % (n + 1), (p // (n + 1)) % (n + 1), p // (n + 1) ** 2


def test_classification_of_a_block(oracle_lib):
    """BPS.C:269-441: the eight block corners sit on edge mesh points with three edges -> corner points; the other points
    of the twelve block edges are within distanceTolerance of an initial edge -> feature edge points; every boundary point
    is a smoothing surface point with the default -smoothingPatches '(".*")' (SM.C:1837-1840)."""
    from smoothmesh_amd.meshgen import hex_block
    n = 6
    m = hex_block(n, jitter=0.2, seed=5)
    init, target, surf = boundary_inputs(3, 2)                 # 3 segments per block edge: mesh points between edge points
    o, _, prm, on = make_pair(m, oracle_lib, init, target, surf, engine=False)
    assert on
    f = o.boundary_fields()
    i, j, k = _ijk(n)
    ext = lambda a: (a == 0) | (a == n)
    n_ext = ext(i).astype(int) + ext(j) + ext(k)
    assert np.array_equal(f["isCornerPoint"].astype(bool), n_ext == 3)
    assert np.array_equal(f["isFeatureEdgePoint"].astype(bool), n_ext == 2)
    assert np.array_equal(f["isSmoothingSurfacePoint"].astype(bool), n_ext >= 1)
    # corner targets = the corners themselves (BPS.C:381-388); one string per block edge (BPS.C:557-587)
    c = f["isCornerPoint"].astype(bool)
    assert np.array_equal(f["cornerPoints"][c], np.array(m.points)[c])
    assert len(np.unique(f["targetEdgeStrings"])) == 12
    # points of one block edge share a string, different edges differ (SM.C:2234-2249)
    e1 = (j == 0) & (k == 0) & (n_ext == 2)
    e2 = (i == 0) & (k == n) & (n_ext == 2)
    s1, s2 = np.unique(f["pointStrings"][e1]), np.unique(f["pointStrings"][e2])
    assert len(s1) == 1 and len(s2) == 1 and s1[0] != s2[0] and s1[0] >= 0
    # OBB.C:396-459: face-interior boundary points hang on the one neighbour a hop inside
    face = (i == 0) & (n_ext == 1)
    assert np.array_equal(f["innerMap"][face], np.arange(len(i))[face] + 1)
    assert np.all(f["innerMap"][n_ext >= 2] == -1)             # edge / corner points have no internal neighbour
    # OBB.C:141-233: inward unit normals
    assert np.array_equal(f["normals"][face], np.tile([1.0, 0.0, 0.0], (face.sum(), 1)))


def test_not_enabled_without_inputs(oracle_lib):
    """SM.C:2080-2093: needs the target surface, the initial edges (or classification lists) and a smoothing patch"""
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(4, jitter=0.1, seed=1)
    init, target, surf = boundary_inputs(4, 2)
    assert not make_pair(m, oracle_lib, init, None, None, engine=False)[3]
    assert not make_pair(m, oracle_lib, None, None, surf, engine=False)[3]
    assert not make_pair(m, oracle_lib, init, None, surf, engine=False, smoothingPatches=())[3]
    assert make_pair(m, oracle_lib, init, None, surf, engine=False, smoothingPatches=("xmin",))[3]


def test_find_line_known_answers(oracle_lib):
    """nearest hit along the segment; misses beyond the segment end and behind its start"""
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(4)
    init, target, surf = boundary_inputs(4, 2)
    o = make_pair(m, oracle_lib, init, None, surf, engine=False)[0]
    hit, p = o.find_line([0.3, 0.4, 0.5], [0.3, 0.4, 2.0])            # from inside through the z = 1 side
    assert hit and np.allclose(p, [0.3, 0.4, 1.0], rtol=0, atol=1e-15)
    hit, p = o.find_line([0.3, 0.4, -1.0], [0.3, 0.4, 2.0])           # crosses z = 0 first, then z = 1
    assert hit and np.allclose(p, [0.3, 0.4, 0.0], rtol=0, atol=1e-15)
    assert not o.find_line([0.3, 0.4, 0.5], [0.3, 0.4, 0.9])[0]       # ends before the surface
    assert not o.find_line([0.3, 0.4, 1.5], [0.3, 0.4, 2.0])[0]       # starts beyond it, pointing away


def test_uniform_block_on_its_own_surface_is_a_fixed_point(oracle_lib):
    """Uniform block, target = the block's own surface: the centroid of a face point's cells lies h/2 inside on the
    point's normal, the ray along the normal brings it back (BPS.C:682-745); edge points are the mean of their two
    surface neighbours' projections, i.e. themselves; corners are pinned -> residual exactly 0."""
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(8, jitter=0.0)                                        # 1/8 is exact in binary
    init, target, surf = boundary_inputs(8, 2)
    o = make_pair(m, oracle_lib, init, None, surf, engine=False)[0]
    n, res, frz = o.iterate(3, 0.0)
    assert n == 3 and np.all(res == 0.0) and np.all(frz == 0)
    assert np.array_equal(o.points(), np.array(m.points))


@pytest.mark.parametrize("constraints", [False, True])
def test_boundary_points_slide_on_the_target(oracle_lib, constraints):
    """jittered block: face points stay in their plane, edge points on their edge, corners fixed, and the tangential
    jitter of the boundary decays (the boundary is smoothed, not frozen)"""
    from smoothmesh_amd.meshgen import hex_block
    n = 8
    m = tangential_jitter(hex_block(n, jitter=0.2, seed=2), 0.03, seed=9)
    p0 = np.array(m.points).copy()
    init, target, surf = boundary_inputs(n, 3)
    o = make_pair(m, oracle_lib, init, None, surf, constraints=constraints, engine=False)[0]
    nd, res, frz = o.iterate(40, 0.0)
    p = o.points()
    i, j, k = _ijk(n)
    for a, idx in enumerate((i, j, k)):
        assert np.all(p[idx == 0, a] == 0.0) and np.all(p[idx == n, a] == 1.0)      # on the box, exactly
    moved = np.abs(p - p0).max(axis=1)
    ext = ((i == 0) | (i == n)).astype(int) + ((j == 0) | (j == n)) + ((k == 0) | (k == n))
    assert np.all(moved[ext == 3] == 0.0) and moved[ext == 2].max() > 1e-3 and moved[ext == 1].max() > 1e-3
    # spacing along one block edge becomes more uniform
    e = np.where((j == 0) & (k == 0))[0]
    assert np.std(np.diff(p[e, 0])) < 0.5 * np.std(np.diff(p0[e, 0]))
    assert res[-1] < 0.3 * res[0]


def test_boundary_moves_onto_a_scaled_target(oracle_lib):
    """target surface, edges and corners = the block scaled by 1.04 about its centre: corners jump to the scaled
    corners (then step-clamped), face points are projected along their normals, until the boundary lies on the target"""
    from smoothmesh_amd.meshgen import hex_block
    n = 6
    m = hex_block(n, jitter=0.1, seed=4)
    init, target, surf = boundary_inputs(n, 3, warp=scale_about_centre(1.04))
    o = make_pair(m, oracle_lib, init, target, surf, engine=False)[0]
    o.iterate(60, 0.0)
    p = o.points()
    i, j, k = _ijk(n)
    lo, hi = 0.5 - 0.52, 0.5 + 0.52
    for a, idx in enumerate((i, j, k)):
        assert np.allclose(p[idx == 0, a], lo, rtol=0, atol=1e-9) and np.allclose(p[idx == n, a], hi, rtol=0, atol=1e-9)


def test_classification_lists_of_a_previous_run_are_used(oracle_lib):
    """SM.C:2066-2077 + BPS.C:344-349: with isCornerPoint / isFeatureEdgePoint data the edge meshes are not consulted"""
    from smoothmesh_amd.meshgen import hex_block
    n = 4
    m = hex_block(n, jitter=0.1, seed=1)
    init, target, surf = boundary_inputs(n, 2)
    f0 = make_pair(m, oracle_lib, init, None, surf, engine=False)[0].boundary_fields()
    cio = f0["isCornerPoint"].astype(np.int32).copy()
    fio = f0["isFeatureEdgePoint"].astype(np.int32).copy()
    victim = np.where(fio == 1)[0][0]
    fio[victim] = 0                                             # the lists win over the geometry
    f1 = make_pair(m, oracle_lib, init, None, surf, engine=False, cornerIO=cio, featureIO=fio)[0].boundary_fields()
    assert f1["isFeatureEdgePoint"][victim] == 0 and f1["isFeatureEdgePoint"].sum() == f0["isFeatureEdgePoint"].sum() - 1
    assert np.array_equal(f1["isCornerPoint"], f0["isCornerPoint"])


def test_castellated_cavity_snaps_to_the_sphere(oracle_lib):
    """polyhedral mesh with a staircase cavity wall, target surface = the sphere the cavity was carved from, only the
    cavity patch smoothed: the wall points end on the (inscribed) triangulated sphere; the outer box stays where it is"""
    from smoothmesh_amd.polymesh import cavity_mesh
    from smoothmesh_amd.surfgen import box_feature_edges, sphere_surface
    m = cavity_mesh(10)
    p0 = np.array(m.points).copy()
    o = make_pair(m, oracle_lib, box_feature_edges(10), None, sphere_surface(levels=3), engine=False, smoothingPatches=("cavity",))[0]
    f = o.boundary_fields()
    wall = f["isSmoothingSurfacePoint"].astype(bool)
    assert wall.sum() > 400 and f["isCornerPoint"].sum() == 8
    r0 = np.linalg.norm(p0[wall] - 0.5, axis=1)
    assert r0.min() < 0.22 and r0.max() > 0.29                        # the staircase
    o.iterate(30, 0.0)
    p = o.points()
    r = np.linalg.norm(p[wall] - 0.5, axis=1)
    assert r.min() > 0.245 and r.max() <= 0.25 + 1e-12                # between the inscribed facets and the sphere
    box = (~wall) & ((p0 == 0.0) | (p0 == 1.0)).any(axis=1)
    assert np.array_equal(p[box], p0[box])                            # frozen surface points are restored (SM.C:2384-2392)


def _ideal_chains(n_points, edges):
    """maximal chains of an edge mesh: edges joined through the points that have exactly two edges"""
    deg = np.bincount(edges.ravel(), minlength=n_points)
    parent = list(range(len(edges)))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    at = [[] for _ in range(n_points)]
    for i, (a, b) in enumerate(edges):
        at[a].append(i)
        at[b].append(i)
    for p in range(n_points):
        if deg[p] == 2:
            parent[find(at[p][0])] = find(at[p][1])
    return np.array([find(i) for i in range(len(edges))])


def _same_partition(a, b):
    fwd, bwd = {}, {}
    return all(fwd.setdefault(x, y) == y and bwd.setdefault(y, x) == x for x, y in zip(a.tolist(), b.tolist()))


def _product_edge_strings(n_points, edges):
    """the product's host routine (csrc/boundary.cpp) through the C-ABI; needs no GPU"""
    import ctypes as C
    from smoothmesh_amd import _ffi
    e = np.ascontiguousarray(edges, np.int32)
    out, n = np.empty(len(e), np.int32), C.c_int32(0)
    rc = _ffi.lib().smgpu_debug_edge_strings(int(n_points), len(e), e.ctypes.data_as(_ffi.c_i32p), out.ctypes.data_as(_ffi.c_i32p), C.byref(n))
    assert rc == 0
    return out, n.value


def test_edge_strings_of_closed_loops_and_branching_meshes(oracle_lib):
    """findEdgeMeshStrings BPS.C:557-587: a closed loop is one string; three polylines meeting in a point are three; the
    product's routine numbers them exactly as the oracle's"""
    loop = np.array([[i, (i + 1) % 7] for i in range(7)], np.int32)
    s, n = oracle_lib.edge_strings(7, loop)
    assert n == 1 and np.all(s == 0)
    star = np.array([[0, 1], [1, 2], [2, 3], [0, 4], [4, 5], [0, 6], [6, 7], [7, 8], [8, 9]], np.int32)     # three arms from point 0
    s, n = oracle_lib.edge_strings(10, star)
    assert n == 3 and _same_partition(s, _ideal_chains(10, star))
    for npts, e in ((7, loop), (10, star)):
        sp, n_p = _product_edge_strings(npts, e)
        so, n_o = oracle_lib.edge_strings(npts, e)
        assert n_p == n_o and np.array_equal(sp, so)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/testcase3/constant/geometry"), reason="reference tree not present")
def test_edge_strings_on_the_reference_test_cases(oracle_lib):
    """The feature edge files of the reference's own test cases.  Where the reference builds strings (the target edge
    mesh, SM.C:2170) the restated recursion yields the maximal chains -- except on testcase3's edge mesh, where it splits
    two chains: the recursion decides whether to descend by pointEdges()[<edge id>] (BPS.C:534,542), a quirk kept as
    written.  The product's routine and the oracle's agree on every file."""
    import glob
    from smoothmesh_amd.surfgen import read_obj_edges
    seen = {}
    for f in sorted(glob.glob("/root/reference/testcase*/constant/geometry/*Edges.obj")):
        pts, e = read_obj_edges(f)
        so, n_o = oracle_lib.edge_strings(len(pts), e)
        sp, n_p = _product_edge_strings(len(pts), e)
        assert n_p == n_o and np.array_equal(sp, so), f
        seen[f.split("reference/")[1]] = (n_o, len(set(_ideal_chains(len(pts), e).tolist())), _same_partition(so, _ideal_chains(len(pts), e)))
    for case in ("testcase4/constant/geometry/targetEdges.obj", "testcase5/constant/geometry/initEdges.obj",
                 "testcase6/constant/geometry/initEdges.obj", "testcase7/constant/geometry/targetEdges.obj",
                 "testcase8/constant/geometry/initEdges.obj"):
        n, ideal, same = seen[case]
        assert same and n == ideal, (case, n, ideal)
    assert seen["testcase3/constant/geometry/initEdges.obj"][:2] == (17, 15)


def test_golden_fixture_boundary(oracle_lib):
    """Regression fixture written by tests/golden/make_golden_boundary.py (oracle output; see that script)."""
    import importlib.util
    import os
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_golden_boundary", os.path.join(here, "golden", "make_golden_boundary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(here, "golden", "hex6_boundary_seed5.npz"))
    m, init, target, surf = mod.case()
    assert np.array_equal(np.array(m.points), g["points0"])
    o = make_pair(m, oracle_lib, init, target, surf, constraints=True, engine=False, blend=0.4)[0]
    f = o.boundary_fields()
    for k in ("isCornerPoint", "isFeatureEdgePoint", "pointStrings", "innerMap"):
        assert np.array_equal(f[k], g[k]), k
    frz_all, res_all = [], []
    for tag, iters in (("1", 1), ("5", 4), ("15", 10)):
        n, res, frz = o.iterate(iters, 0.0)
        frz_all.append(frz); res_all.append(res)
        assert np.max(np.abs(o.points() - g["points" + tag])) <= 1e-14
    assert np.array_equal(np.concatenate(frz_all), g["nFrozen"])
    assert np.allclose(np.concatenate(res_all), g["residual"], rtol=1e-12, atol=0)


def _multi_boundary_case(oracle_lib, grid, nloc, jitter, constraints, blend=0.3, seed=3, layerPatches=()):
    from smoothmesh_amd import default_params, patch_arrays
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    world = grid[0] * grid[1] * grid[2]
    subs = [hex_subdomain(nloc, grid, r, jitter=jitter, seed=seed) for r in range(world)]
    hi = tuple(float(g) for g in grid)                      # every sub-domain is a unit cube
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    pa = [patch_arrays(s.mesh, layerPatches) + (patch_arrays(s.mesh, ('".*"',))[3],) for s in subs]
    on = mo.setup_boundary(pa, (0.3, prm.minEdgeLength, 1.3, 1, 4), box_feature_edges(8, hi=hi), None, box_surface(4, hi=hi), blend)
    assert on
    mo.params = prm
    return mo, orcs, subs, (off, dom, loc), hi


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])
def test_decomposed_boundary_smoothing_keeps_the_ranks_consistent(oracle_lib, grid):
    """Under -parallel every rank projects its copy of a shared boundary point itself; the reference's syncs (normals
    OBB.C:184-198, inner neighbour coordinates OBB.C:490-496, feature projections BPS.C:659-674, centroid sums SM.C:134-148)
    give all sharers the same inputs, so the copies must stay identical; the boundary stays on the block"""
    mo, orcs, subs, (off, dom, loc), hi = _multi_boundary_case(oracle_lib, grid, (4, 5, 6), 0.25, True)
    n, res, frz = mo.iterate(10, 0.0)
    assert n == 10 and np.isfinite(res).all() and res[-1] < 0.5 * res[0]
    P = [o.points() for o in orcs]
    for i in range(len(off) - 1):
        c = [P[dom[k]][loc[k]] for k in range(off[i], off[i + 1])]
        assert all(np.array_equal(c[0], x) for x in c[1:])
    allp = np.concatenate(P)
    assert np.array_equal(allp.min(axis=0), [0.0, 0.0, 0.0]) and np.array_equal(allp.max(axis=0), hi)
    # a uniform decomposed block on its own surface is a fixed point, as in the serial run
    mo2, orcs2, subs2, _, _ = _multi_boundary_case(oracle_lib, grid, (4, 4, 4), 0.0, False)      # 1/4 is exact in binary
    n, res, frz = mo2.iterate(3, 0.0)
    assert np.all(res == 0.0) and np.all(frz == 0)
    for o, s in zip(orcs2, subs2):
        assert np.array_equal(o.points(), np.array(s.mesh.points))


def test_find_line_on_a_tilted_triangle(oracle_lib):
    """analytic hit: the plane x + y + z = 1 cut by the segment from the origin towards (1, 1, 1) at t = 1/3; a segment
    that stops short misses, one that passes outside the triangle misses"""
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.surfgen import box_feature_edges
    m = hex_block(2)
    tri = (np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]), np.array([[0, 1, 2]], np.int32))
    o = make_pair(m, oracle_lib, box_feature_edges(2), None, tri, engine=False)[0]
    hit, p = o.find_line([0, 0, 0], [1, 1, 1])
    assert hit and np.allclose(p, [1 / 3, 1 / 3, 1 / 3], rtol=0, atol=1e-15)
    assert not o.find_line([0, 0, 0], [0.3, 0.3, 0.3])[0]
    assert not o.find_line([2, 2, -1], [2, 2, 3])[0]
    hit, p = o.find_line([0.2, 0.2, -1], [0.2, 0.2, 2])                   # vertical through (0.2, 0.2, 0.6)
    assert hit and np.allclose(p, [0.2, 0.2, 0.6], rtol=0, atol=1e-15)


def test_feature_edge_point_moves_to_the_mean_projection_of_its_surface_neighbours(oracle_lib):
    """BPS.C:623-677 + 883-893 + SM.C:2356: a point on a block edge has two surface neighbours (one on each adjacent side);
    its target is the mean of their projections onto the edge, and the step clamp (relStepFrac 0.5, step below maxStepLength)
    takes it half of the way"""
    from smoothmesh_amd.meshgen import hex_block
    n = 6
    m = hex_block(n, jitter=0.0)
    p0 = np.array(m.points).copy()
    i, j, k = _ijk(n)
    idx = lambda a, b, c: a + (n + 1) * (b + (n + 1) * c)
    e, q1, q2 = idx(3, 0, 0), idx(3, 1, 0), idx(3, 0, 1)                  # edge point on the x axis and its two surface neighbours
    p0[q1, 0] += 0.02
    p0[q2, 0] += 0.05
    m.points[:] = p0
    init, target, surf = boundary_inputs(n, 2)
    o = make_pair(m, oracle_lib, init, None, surf, engine=False, maxStepLength=10.0)[0]
    o.iterate(1, 0.0)
    p = o.points()
    want = p0[e, 0] + 0.5 * (0.5 * (p0[q1, 0] + p0[q2, 0]) - p0[e, 0])
    assert abs(p[e, 0] - want) <= 1e-15 and p[e, 1] == 0.0 and p[e, 2] == 0.0
