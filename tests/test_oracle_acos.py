"""acos on the path (SM.C:782-783, 992-995): the engine evaluates it as a fixed sequence of IEEE operations (csrc/smacos.hpp), the
oracle with glibc's std::acos (the reference's arithmetic) or -- on request -- with that same sequence.  Here, on the CPU: the
sequence is accurate, its special values are right, and over whole constrained runs the two variants lead to the SAME decisions
(frozen sets, coordinates bit for bit) although their angles differ in the last bit for ~6 % of the arguments; a census of the
threshold comparisons says how close any of them came to being decided by that last bit."""
import math

import numpy as np
import pytest

from oracle import oracle_ffi


def test_device_acos_is_accurate_and_hits_the_special_values():
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-1, 1, 120000), np.linspace(-0.99999, 0.99999, 40001),
                         [0.5, -0.5, np.nextafter(0.5, 0), np.nextafter(0.5, 1), -np.nextafter(0.5, 0), 0.0, 1e-300, -1e-300, 0.99999, -0.99999]])
    ref = np.arccos(xs.astype(np.longdouble))                      # 64-bit mantissa: 11 bits beyond a double
    dev = np.array([oracle_ffi.acos(x, "device") for x in xs])
    gl = np.array([oracle_ffi.acos(x, "glibc") for x in xs])
    ulp = np.spacing(np.abs(ref).astype(np.float64))
    err_dev = float(np.max(np.abs(dev.astype(np.longdouble) - ref) / ulp))
    err_gl = float(np.max(np.abs(gl.astype(np.longdouble) - ref) / ulp))
    assert err_dev < 1.0 and err_gl < 1.0, (err_dev, err_gl)
    assert float(np.max(np.abs(dev - gl) / ulp)) <= 1.0            # never more than the last bit apart ...
    assert 0 < np.count_nonzero(dev != gl) < 0.15 * len(xs)        # ... and that for a minority of the arguments
    assert oracle_ffi.acos(1.0) == 0.0 and oracle_ffi.acos(-1.0) == math.pi and oracle_ffi.acos(0.0) == math.pi / 2
    assert math.isnan(oracle_ffi.acos(float("nan")))
    # monotone over the clamped range of the path
    g = np.linspace(-0.99999, 0.99999, 20001)
    a = np.array([oracle_ffi.acos(x) for x in g])
    assert np.all(np.diff(a) <= 0)
    # the clamp bounds of the reference (acos(+-0.99999), SURVEY 8c: 0.004472 / 3.137121), to the last bit of glibc's
    for c in (0.99999, -0.99999):
        assert abs(oracle_ffi.acos(c) - math.acos(c)) <= np.spacing(math.acos(c))
    assert abs(oracle_ffi.acos(0.99999) - 0.004472) < 1e-6 and abs(oracle_ffi.acos(-0.99999) - 3.137121) < 1e-6


def _cases():
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import cavity_mesh
    return [("hex 12x10x9, jitter 0.35", hex_block(12, 10, 9, jitter=0.35, seed=4), dict(minAngle=50.0, maxAngle=125.0), 8),
            ("hex 10^3, jitter 0.3, thresholds at the block's own right angles", hex_block(10, 10, 10, jitter=0.3, seed=9), dict(minAngle=88.0, maxAngle=92.0), 5),
            ("polyhedral cavity mesh 14^3", cavity_mesh(14, jitter=0.25, seed=6), dict(minAngle=35.0, maxAngle=160.0), 6)]


@pytest.mark.parametrize("k", [0, 1, 2])
def test_both_acos_variants_lead_to_the_same_decisions(oracle_lib, k):
    from smoothmesh_amd import default_params
    name, mesh, over, iters = _cases()[k]
    out = {}
    prev = oracle_ffi.set_acos_variant("glibc")
    try:
        for variant in ("glibc", "device"):
            oracle_ffi.set_acos_variant(variant)
            o = oracle_lib.Oracle(mesh)
            o.set_params(default_params(o.mesh_stats()[0], edgeAngleConstraint=True, faceAngleConstraint=True, **over))
            oracle_ffi.acos_census(True)
            n, res, frz = o.iterate(iters, 0.0)
            census = oracle_ffi.acos_census(False)
            o.phaseA(); o.phaseB()
            out[variant] = (n, res.copy(), frz.copy(), o.points().copy(), {f: o.field(f).copy() for f in ("pointMinAngle", "pointMaxAngle", "eaMinC", "eaMinN")}, census)
            o.close()
    finally:
        oracle_ffi.set_acos_variant(prev)
    a, b = out["glibc"], out["device"]
    assert a[0] == b[0] == iters and frozen_busy(a[2]), (name, a[2])
    assert np.array_equal(a[2], b[2]), name                                # the same points frozen in every iteration
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[1], b[1]), name  # hence the same coordinates and residuals, bit for bit
    for f in a[4]:                                                          # the angles themselves: the last bit at most
        x, y = a[4][f], b[4][f]
        fin = np.isfinite(x) & np.isfinite(y) & (np.abs(x) < 10)
        assert np.array_equal(fin, np.isfinite(y) & (np.abs(y) < 10))
        assert np.max(np.abs(x[fin] - y[fin]) / np.spacing(np.maximum(np.abs(x[fin]), 1e-3))) <= 2.0, (name, f)
    # how close did a comparison come to being decided by the last bits?  (sides with EQUAL bits are the same function of the same
    # inputs -- a point that did not move -- and are counted apart)
    for variant in ("glibc", "device"):
        c = out[variant][5]
        assert c["comparisons"] > 1000 and c["within_8ulp"] == 0, (name, variant, c)
        print(f"{name} [{variant}]: {c['comparisons']} threshold comparisons, {c['equal']} with equal sides, none within 8 ulp; closest unequal pair {c['min_ulp']} ulp apart")


def frozen_busy(frz):
    return int(np.max(frz)) > 0


def test_near_tie_classes_of_the_census(oracle_lib):
    """the engine's near-tie census (include/smgpu.h: smgpu_iter_stats::nNearTies, smgpu_get_near_ties) counted on the oracle's side:
    by comparison (SM.C:923 / 1367 / walk verdicts), sides 1 .. window ulp apart.  With the default window (4 ulp) nothing on these
    meshes; a window as wide as the doubles catches every comparison with unequal sides, class by class"""
    from smoothmesh_amd import default_params
    name, mesh, over, iters = _cases()[1]            # the block with the thresholds at its own right angles
    o = oracle_lib.Oracle(mesh)
    o.set_params(default_params(o.mesh_stats()[0], edgeAngleConstraint=True, faceAngleConstraint=True, **over))
    try:
        oracle_ffi.acos_census_window(4)
        oracle_ffi.acos_census(True)
        o.iterate(2, 0.0)
        c4 = oracle_ffi.acos_census(False)
        assert c4["near"] == {"edge_angle": 0, "good_range": 0, "walk": 0} and c4["within_8ulp"] == 0
        o.set_points(mesh.points)
        oracle_ffi.acos_census_window(2 ** 62)
        oracle_ffi.acos_census(True)
        o.iterate(2, 0.0)
        cw = oracle_ffi.acos_census(False)
    finally:
        oracle_ffi.acos_census_window(4)
        o.close()
    near = cw["near"]
    assert near["edge_angle"] > mesh.nPoints // 2 and near["good_range"] >= mesh.nPoints and near["walk"] > 0
    # every comparison with unequal sides is in exactly one class, except re-visits of a point by the walk (SM.C:1367 again)
    assert sum(near.values()) <= cw["comparisons"] - cw["equal"]
    assert near["good_range"] <= 2 * 2 * mesh.nPoints          # at most two comparisons per point and iteration
