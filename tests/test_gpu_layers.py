"""GPU parity of the optional boundary layer treatment (include/smgpu.h smgpu_set_layers; reference
SM.C:2186-2221 set-up, SM.C:2266 + 2283-2305 per iteration) against the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu


def _pair(mesh, oracle_lib, layerPatches, constraints, **lp):
    from smoothmesh_amd import LayerParams, SmoothEngine, default_params, patch_arrays
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    prm = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(prm)
    e.set_params(prm)
    L = LayerParams(layerPatches=tuple(layerPatches), **lp)
    st, sz, kd, sel = patch_arrays(mesh, L.layerPatches)
    on_o = o.setup_layers(st, sz, kd, sel, L.layerMaxBlendingFraction,
                          prm.minEdgeLength if L.layerEdgeLength is None else L.layerEdgeLength, L.layerExpansionRatio,
                          L.minLayers, L.maxLayers)
    on_g = e.set_layers(L, prm.minEdgeLength)
    assert on_o == on_g
    return o, e, on_g


def _check_setup(o, e):
    f = o.layer_fields()
    assert np.array_equal(e.debug_field("layerHops").astype(np.int32), f["hops"])
    assert np.array_equal(e.debug_field("layerOuterMap").astype(np.int32), f["outerMap"])
    assert np.array_equal(e.debug_field("layerNormals").reshape(-1, 3), f["normals"])     # same operations: same bits


@pytest.mark.parametrize("patches,constraints", [(["xmin"], False), (["xmin", "ymax"], False), (['"z.*"', "xmax"], True)])
def test_hex_block_with_layers(oracle_lib, patches, constraints):
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(12, 10, 9, jitter=0.25, seed=11)
    o, e, on = _pair(m, oracle_lib, patches, constraints, layerExpansionRatio=1.2)
    assert on
    _check_setup(o, e)
    n_o, res_o, frz_o = o.iterate(10, 0.0)
    n_g, res_g, frz_g = e.iterate(10, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    assert rel_linf(e.get_points(), o.points()) <= 1e-13
    # the treatment is really active: the result differs from a run without it
    o2, e2, _ = _pair(m, oracle_lib, [], constraints)
    e2.iterate(10, 0.0)
    assert rel_linf(e2.get_points(), e.get_points()) > 1e-6


@pytest.mark.parametrize("constraints", [False, True])
def test_polyhedral_cavity_wall_layers(oracle_lib, constraints):
    """castellated cavity surface: concave steps give multiply connected wall points (OBB.C:311-320)"""
    from smoothmesh_amd.polymesh import cavity_mesh
    m = cavity_mesh(14)
    o, e, on = _pair(m, oracle_lib, ["cavity"], constraints, maxLayers=3, layerMaxBlendingFraction=0.4)
    assert on
    _check_setup(o, e)
    n_o, res_o, frz_o = o.iterate(6, 0.0)
    n_g, res_g, frz_g = e.iterate(6, 0.0)
    assert np.array_equal(frz_o, frz_g)
    assert rel_linf(e.get_points(), o.points()) <= 1e-13


def test_tiled_and_direct_kernels_agree_with_layers(oracle_lib, monkeypatch):
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(9, 8, 7, jitter=0.3, seed=5)
    o, e, _ = _pair(m, oracle_lib, ["ymin"], False)
    e.iterate(5, 0.0)
    monkeypatch.setenv("SMGPU_TILES", "0")
    o2, e2, _ = _pair(m, oracle_lib, ["ymin"], False)
    e2.iterate(5, 0.0)
    assert np.array_equal(e.get_points(), e2.get_points())
    o.iterate(5, 0.0)
    assert rel_linf(e.get_points(), o.points()) <= 1e-13


def test_disabled_and_refused_cases(oracle_lib):
    from smoothmesh_amd import LayerParams, SmgpuError, SmoothEngine, default_params
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(5, jitter=0.1, seed=2)
    e = SmoothEngine(m)
    prm = default_params(e.mesh_stats()[0])
    e.set_params(prm)
    assert e.set_layers(LayerParams(layerPatches=("nosuchpatch",)), prm.minEdgeLength) is False       # SM.C:2025
    assert e.set_layers(LayerParams(layerPatches=("xmin",), layerMaxBlendingFraction=0.0), prm.minEdgeLength) is False
    with pytest.raises(SmgpuError):
        e.set_layers(LayerParams(layerPatches=("xmin",), maxLayers=-2), prm.minEdgeLength)


@pytest.mark.parametrize("grid,patches,constraints,sub", [((2, 1, 1), ["xmin"], False, (5, 4, 4)), ((2, 2, 1), ["xmin", "zmax"], False, (5, 4, 4)),
                                                          ((2, 2, 2), ['"x.*"', "ymin"], True, (4, 4, 3)), ((2, 2, 1), ["xmin", "ymax"], False, (14, 12, 10))])
def test_layers_under_parallel_match_the_multi_domain_oracle(oracle_lib, grid, patches, constraints, sub):
    """-parallel + -layerPatches: step-wise set-up with the host doing the reference's syncPointList calls, the 6-double
    layer exchange next to exchange A; expected = the oracle's MultiDomain.  The last case spans several tiles and a
    point shared by four ranks that hangs on different neighbours on different ranks."""
    from smoothmesh_amd import LayerParams, default_params, patch_arrays
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.halo import LocalMultiSmoother
    from smoothmesh_amd.meshgen import hex_subdomain
    world = grid[0] * grid[1] * grid[2]
    subs = [hex_subdomain(sub, grid, r, jitter=0.25, seed=13) for r in range(world)]
    ms = LocalMultiSmoother(subs, device=0)
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    mn = min(o.mesh_stats()[0] for o in orcs)
    prm = default_params(mn, edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    ms.set_params(prm)
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    lp = LayerParams(layerPatches=tuple(patches), layerExpansionRatio=1.2)
    assert mo.setup_layers([patch_arrays(s.mesh, patches) for s in subs], lp.layerMaxBlendingFraction, prm.minEdgeLength,
                           lp.layerExpansionRatio, lp.minLayers, lp.maxLayers)
    assert ms.set_layers(lp, prm.minEdgeLength)
    for st, o in zip(ms.states, orcs):
        f = o.layer_fields()
        assert np.array_equal(st.eng.debug_field("layerHops").astype(np.int32), f["hops"])
        assert np.array_equal(st.eng.debug_field("layerOuterMap").astype(np.int32), f["outerMap"])
        assert np.array_equal(st.eng.debug_field("layerNormals").reshape(-1, 3), f["normals"])
    n_o, res_o, frz_o = mo.iterate(8, 0.0)
    n_g, res_g, frz_g = ms.iterate(8, 0.0)
    assert np.array_equal(frz_o, frz_g)
    for o, pts in zip(orcs, ms.get_points()):
        assert rel_linf(pts, o.points()) <= 1e-13
