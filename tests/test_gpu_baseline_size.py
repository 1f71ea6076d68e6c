"""The HIP path against the CPU oracle ON THE BASELINE MESHES THEMSELVES (BASELINE.json configs[1..3]).

The small-mesh parity tests cannot reach the code that only runs big: tile tables with thousands of tiles, the XCD-contiguous
launch with a tile count that is no multiple of 8, ELL rows past the two pre-loaded chunks, the scans over 40 k block counts,
the list compaction of the exact face-angle pass, the fixed-point walk on percolated components.  Here `iterate(k)` runs on both
sides from the same coordinates (the loop body SM.C:2257-2437) and the nFrozenPoints series must be identical and the
coordinates within 1e-13 relative L-inf (north star: 1e-10; measured: bit-equal).

Cost: the serial oracle takes ~0.25 / 1.3 / 1.7 s per iteration on the 1 M-cell meshes and ~50 s of set-up plus ~30 s per
constrained iteration on the 10 M-cell mesh -- k is sized accordingly.
"""
import os

import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu

COORD_TOL = 1e-13


@pytest.fixture(scope="module")
def cavity215():
    from smoothmesh_amd.polymesh import cavity_mesh
    return cavity_mesh(215, jitter=0.2, seed=12345)


def _compare(mesh, oracle_lib, k, check_every=None, **over):
    from smoothmesh_amd import SmoothEngine, default_params
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    mn_o, mx_o = o.mesh_stats()
    assert (mn_o, mx_o) == e.mesh_stats()
    p = default_params(mn_o, **over)
    o.set_params(p)
    e.set_params(p)
    try:
        done = 0
        frz_o_all, frz_g_all = [], []
        for chunk in (check_every or [k]):
            n_o, res_o, frz_o = o.iterate(chunk, 0.0)
            n_g, res_g, frz_g = e.iterate(chunk, 0.0)
            done += chunk
            assert n_o == n_g == chunk
            assert np.array_equal(frz_o, frz_g), (done, frz_o, frz_g)
            assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
            err = rel_linf(e.get_points(), o.points())
            assert err <= COORD_TOL, (done, err)
            frz_o_all.append(frz_o); frz_g_all.append(frz_g)
        assert done == k
        return np.concatenate(frz_g_all), bool(np.array_equal(e.get_points(), o.points()))
    finally:
        e.close()
        o.close()


def test_hex100_constraints_off_matches_oracle(oracle_lib):
    """configs[1]: 100^3 hex block, constraints off -- 5 iterations, compared after 2 and after 5"""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(100, jitter=0.2, seed=12345)
    frz, bitwise = _compare(mesh, oracle_lib, 5, check_every=[2, 3], edgeAngleConstraint=False, faceAngleConstraint=False)
    assert np.all(frz == mesh.nPoints - 99 ** 3)      # only the boundary points count as frozen
    assert bitwise


def test_hex100c_constraints_on_matches_oracle(oracle_lib):
    """configs[2]: the same block with the edge- and face-angle constraints on (minAngle 35 / maxAngle 160)"""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(100, jitter=0.2, seed=12345)
    frz, bitwise = _compare(mesh, oracle_lib, 5, check_every=[1, 4])
    assert bitwise


def test_hex100c_busy_constraints_match_oracle(oracle_lib):
    """the same size with heavier jitter and minAngle 60: thousands of points are frozen by the evaluators and the walk, so
    the filters' UNSURE lists, the list-based exact pass and the walk replay are busy at a size with ~4 000 tiles"""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(100, jitter=0.42, seed=12345)
    frz, _ = _compare(mesh, oracle_lib, 4, check_every=[1, 3], minAngle=60.0)
    assert frz[0] > mesh.nPoints - 99 ** 3 + 1000


@pytest.mark.parametrize("constraints", [False, True])
def test_cavity100_matches_oracle(oracle_lib, constraints):
    """1 M-cell castellated polyhedral cavity mesh (the 10 M-cell configs[3] family at a size the oracle runs in seconds):
    mixed tiles (general ELL loops next to the unrolled hex paths), the refinement interface keeps the walk busy"""
    from smoothmesh_amd.polymesh import cavity_mesh
    mesh = cavity_mesh(100, jitter=0.2, seed=12345)
    frz, bitwise = _compare(mesh, oracle_lib, 4, check_every=[1, 3], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    if constraints:
        assert frz[-1] > frz[0]          # the walk freezes more and more points along the interface
    assert bitwise


def test_cavity215c_matches_oracle(oracle_lib, cavity215):
    """configs[3] itself: the 10 M-cell polyhedral mesh, constraints on -- TWELVE iterations on both sides, compared after 1, 2, 6
    and 12 (the components of the freeze walk's interaction graph start tiny and percolate within ten iterations: the causal
    fixed-point replay, its warm start from the previous iteration and the mid-run change of the replay form are checked against
    the oracle at configs[3]'s own size, not against the host replay), then one more with the constraints off on the same
    engine / oracle pair (parameters changed mid-run: the fused kernel path at 40 k tiles).  The oracle costs ~50 s of set-up and
    ~15-30 s per constrained iteration."""
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = cavity215
    assert mesh.nCells > 9_500_000
    o = oracle_lib.Oracle(mesh)
    e = SmoothEngine(mesh)
    try:
        p = default_params(o.mesh_stats()[0])
        o.set_params(p); e.set_params(p)
        done, series = 0, []
        # (SMOOTHMESH_BIG_TESTS=1: twelve iterations as in rounds 3-4; the default selection stops at eight -- the serial oracle is
        # ~20 s per constrained iteration at this size, and the whole GPU suite has to stay well inside the driver's time limit;
        # BASELINE's full 200 iterations of this configuration are compared in profiles/r4/parity_long.jsonl)
        big = bool(os.environ.get("SMOOTHMESH_BIG_TESTS"))
        for chunk in ((1, 1, 4, 6) if big else (1, 1, 2, 4)):
            n_o, res_o, frz_o = o.iterate(chunk, 0.0)
            n_g, res_g, frz_g = e.iterate(chunk, 0.0)
            done += chunk
            assert n_o == n_g == chunk
            assert np.array_equal(frz_o, frz_g), (done, frz_o, frz_g)
            assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
            assert rel_linf(e.get_points(), o.points()) <= COORD_TOL, done
            series += [int(x) for x in frz_g]
        assert done == (12 if big else 8) and series[0] > 500_000 and series[-1] != series[0]
        assert np.array_equal(e.get_points(), o.points())                     # measured: bit-equal
        p2 = default_params(o.mesh_stats()[0], edgeAngleConstraint=False, faceAngleConstraint=False)
        o.set_params(p2); e.set_params(p2)
        n_o, res_o, frz_o = o.iterate(1, 0.0)
        n_g, res_g, frz_g = e.iterate(1, 0.0)
        assert np.array_equal(frz_o, frz_g)
        assert np.array_equal(e.get_points(), o.points())
    finally:
        e.close()
        o.close()


def test_walk_replay_places_agree_on_the_10M_cell_mesh(monkeypatch, cavity215):
    """cavity215c: the fixed-point device replay against the host replay (SMGPU_WALK=host) at the size of configs[3] --
    identical nFrozenPoints series, residuals and coordinates after 6 iterations (the components start to merge)"""
    from smoothmesh_amd import SmoothEngine, default_params
    mesh = cavity215
    outs = {}
    for walk in ("fix", "host"):
        monkeypatch.setenv("SMGPU_WALK", walk)
        e = SmoothEngine(mesh)
        e.set_params(default_params(e.mesh_stats()[0]))
        n, res, frz = e.iterate(6, 0.0)
        outs[walk] = (res, frz, e.get_points())
        e.close()
    assert np.array_equal(outs["fix"][1], outs["host"][1])
    assert np.array_equal(outs["fix"][0], outs["host"][0])
    assert np.array_equal(outs["fix"][2], outs["host"][2])


BIG = pytest.mark.skipif(not __import__("os").environ.get("SMOOTHMESH_BIG_TESTS"),
                         reason="tens of GiB on host and device and minutes of serial oracle: set SMOOTHMESH_BIG_TESTS=1 "
                                "(last run: profiles/r5/big_mesh_parity.txt)")


@BIG
def test_hex400_64M_cells_matches_oracle(oracle_lib):
    """BEYOND the baseline sizes: a 64 M-cell block (82 GiB on the device, index products past 2^29), constraints off -- two
    iterations, bit for bit.  The sizes at which 32-bit index arithmetic would first go wrong are only reachable this way."""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(400, jitter=0.2, seed=12345)
    frz, bitwise = _compare(mesh, oracle_lib, 2, check_every=[1, 1], edgeAngleConstraint=False, faceAngleConstraint=False)
    assert np.all(frz == mesh.nPoints - 399 ** 3)
    assert bitwise


@BIG
def test_cavity300c_26M_cells_matches_oracle(oracle_lib):
    """the polyhedral mesh at 2.6 x configs[3]'s size, constraints on: two iterations, bit for bit"""
    from smoothmesh_amd.polymesh import cavity_mesh
    mesh = cavity_mesh(300, jitter=0.2, seed=12345)
    assert mesh.nCells > 26_000_000
    frz, bitwise = _compare(mesh, oracle_lib, 2, check_every=[1, 1])
    assert bitwise
