"""GPU test of the device side of the multi-rank path on ONE GPU: every sub-domain gets its own engine
(handle) on the same device and the exchange is a device-side copy (LocalMultiSmoother); expected =
the oracle's MultiDomain (the reference under mpirun with the same decomposition)."""
import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu


# the (16, 12, 12) sub-domains span several geometry/smoothing tiles, so the interior/shared tile split and the
# look-ahead geometry (smgpu_iter_ahead) are really exercised
# overlap = 1: engines compute on their own streams, the exchange runs on torch's stream (smgpu_halo_desc.exchangeStream)
@pytest.mark.parametrize("grid,constraints,sub,overlap", [
    ((2, 1, 1), False, (5, 4, 4), 0), ((2, 1, 1), True, (5, 4, 4), 0), ((2, 2, 2), True, (5, 4, 4), 1),
    ((3, 1, 2), True, (5, 4, 4), 0), ((2, 2, 1), False, (16, 12, 12), 0), ((2, 2, 1), False, (16, 12, 12), 1),
    ((2, 1, 2), True, (16, 12, 12), 0), ((2, 1, 2), True, (16, 12, 12), 1)])
def test_local_multi_smoother_matches_multi_oracle(oracle_lib, grid, constraints, sub, overlap):
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.halo import LocalMultiSmoother
    from smoothmesh_amd.meshgen import hex_subdomain
    world = grid[0] * grid[1] * grid[2]
    subs = [hex_subdomain(sub, grid, r, jitter=0.3, seed=9) for r in range(world)]
    ms = LocalMultiSmoother(subs, device=0, overlap=bool(overlap))
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    mn = min(o.mesh_stats()[0] for o in orcs)
    assert mn == ms.global_min_edge()
    prm = default_params(mn, edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    ms.set_params(prm)
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    n_o, res_o, frz_o = mo.iterate(8, 0.0)
    big_fused = sub[0] > 8 and not constraints
    if big_fused:
        ms.states[0].eng.enable_timing(True)
    n_g, res_g, frz_g = ms.iterate(8, 0.0)
    if big_fused:
        # with an exchange stream the look-ahead is active: iteration 1 = one full geometry launch, every later one =
        # shared tiles, and every iteration launches the look-ahead for the interior tiles -> 1 + 7 + 8 launches;
        # in order (no exchange stream) nothing would overlap, so the launches are not split: 8
        geom = [c for c in ms.states[0].eng.counters() if c["name"] == "k_geom_tile"]
        assert geom and geom[0]["launches"] == (16 if overlap else 8)
    assert n_o == n_g
    assert np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    for o, pts in zip(orcs, ms.get_points()):
        assert rel_linf(pts, o.points()) <= 1e-13
    # duplicated (shared) points stay bit-identical across the engines that hold them
    g = np.concatenate([s.pointProcAddressing for s in subs])
    allp = np.concatenate(ms.get_points())
    order = np.argsort(g, kind="stable")
    gs, ps = g[order], allp[order]
    same = gs[1:] == gs[:-1]
    assert np.array_equal(ps[1:][same], ps[:-1][same])


def _poly_case(oracle_lib, N, grid, constraints, jitter=0.2, seed=4):
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.polymesh import cavity_subdomain
    world = grid[0] * grid[1] * grid[2]
    subs = [cavity_subdomain(N, grid, r, jitter=jitter, seed=seed) for r in range(world)]
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    for o in orcs:
        o.set_params(prm)
    table = shared_point_table(subs)
    return subs, orcs, prm, table, oracle_lib.MultiOracle(orcs, *table)


# BASELINE configs[4]'s workload in small: the castellated POLYHEDRAL cavity mesh cut into boxes by planes that pass through the
# refinement shell -- irregular shared sets, hanging-node (5..8-vertex) faces on processor patches, reversed processor
# faces, refinement-interface points shared by four ranks (the centre line).  N = 20: the sub-domains span several tiles.
@pytest.mark.parametrize("N,grid,constraints,overlap", [
    (10, (2, 1, 1), False, 0), (10, (2, 1, 1), True, 0), (12, (2, 2, 1), False, 1), (12, (2, 2, 1), True, 0),
    (12, (2, 2, 2), False, 0), (12, (2, 2, 2), True, 1), (11, (3, 1, 2), True, 0), (20, (2, 2, 2), False, 1), (20, (2, 1, 2), True, 0)])
def test_polyhedral_decomposition_matches_multi_oracle(oracle_lib, N, grid, constraints, overlap):
    from smoothmesh_amd.halo import LocalMultiSmoother
    subs, orcs, prm, (off, dom, loc), mo = _poly_case(oracle_lib, N, grid, constraints)
    world = len(subs)
    if world >= 4:
        # a point with >= 3 sharers that lies on a polygonal (hanging-node) face: a refinement-interface point on a processor patch
        found = False
        for i in np.flatnonzero(np.diff(off) >= 3):
            d, l = int(dom[off[i]]), int(loc[off[i]])
            m = subs[d].mesh
            faces = np.searchsorted(m.faceOffsets, np.flatnonzero(m.facePoints == l), side="right") - 1
            if np.any(np.diff(m.faceOffsets)[faces] > 4):
                found = True
                break
        assert found
    ms = LocalMultiSmoother(subs, device=0, overlap=bool(overlap))
    assert ms.global_min_edge() == min(o.mesh_stats()[0] for o in orcs)
    ms.set_params(prm)
    n_o, res_o, frz_o = mo.iterate(7, 0.0)
    n_g, res_g, frz_g = ms.iterate(7, 0.0)
    assert n_o == n_g
    assert np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    for o, pts in zip(orcs, ms.get_points()):
        assert rel_linf(pts, o.points()) <= 1e-13
    if constraints:
        assert frz_g[-1] > sum(int((~s.mesh.find_internal_points().astype(bool)).sum()) for s in subs)   # the constraints did freeze points
    # duplicated (shared) points stay bit-identical across the engines that hold them
    g = np.concatenate([s.pointProcAddressing for s in subs])
    allp = np.concatenate(ms.get_points())
    order = np.argsort(g, kind="stable")
    gs, ps = g[order], allp[order]
    same = gs[1:] == gs[:-1]
    assert same.any() and np.array_equal(ps[1:][same], ps[:-1][same])


def test_inline_pack_knob_gives_the_same_result(oracle_lib, monkeypatch):
    """SMGPU_HALO_INLINE_PACKF=1: the shared points' freeze flags packed inside the smoothing kernel (no k_halo_packF launch)"""
    from smoothmesh_amd.halo import LocalMultiSmoother
    monkeypatch.setenv("SMGPU_HALO_INLINE_PACKF", "1")
    subs, orcs, prm, table, mo = _poly_case(oracle_lib, 12, (2, 2, 2), False)
    ms = LocalMultiSmoother(subs, device=0, overlap=False)
    ms.set_params(prm)
    n_o, res_o, frz_o = mo.iterate(6, 0.0)
    n_g, res_g, frz_g = ms.iterate(6, 0.0)
    assert np.array_equal(frz_o, frz_g)
    for o, pts in zip(orcs, ms.get_points()):
        assert rel_linf(pts, o.points()) <= 1e-13


def test_distributed_smoother_polyhedral_two_processes():
    """One process per rank (torch.distributed.run) with the real engines, two ranks sharing this box's GPU through the gloo
    debug transport: DistributedSmoother on the two halves of the polyhedral cavity mesh (each rank generates its own)
    equals the oracle's MultiDomain bit for bit, constraints off and on, in order and overlapped (scripts/check_dist_poly.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMOOTHMESH_SHARE_GPU="1", SMOOTHMESH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29523", os.path.join(root, "scripts", "check_dist_poly.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ok ") == 8 and "BAD" not in r.stdout


def test_peer_store_transport_two_processes():
    """SMOOTHMESH_EXCHANGE=push: the ranks' kernels store the shared-point records into each other's receive slots (buffers
    mapped across the two PROCESSES with hipIpc; here both live on this box's one GPU) and signal with flag words; the host
    exchanges nothing per iteration.  Polyhedral two-way decomposition against the oracle's MultiDomain, constraints off and on
    (exchange A and F), and the boundary smoothing + layer case (exchange L rides with A; the L records grow from 6 to 14
    doubles when the boundary set-up follows the layer set-up, so the destination tables are rebuilt)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMOOTHMESH_SHARE_GPU="1", SMOOTHMESH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SMOOTHMESH_EXCHANGE="push")
    for script, port, marks in (("check_dist_poly.py", "29531", 8), ("check_dist_boundary.py", "29533", None)):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", port, os.path.join(root, "scripts", script)], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, script + r.stdout[-3000:] + r.stderr[-3000:]
        assert "BAD" not in r.stdout
        if marks:
            assert r.stdout.count(": ok ") == marks and r.stdout.count("[peer stores") == marks


@pytest.mark.parametrize("overlap", [False, True])
def test_single_rank_distributed_history_equals_serial(oracle_lib, overlap):
    """A rank without shared points launches no geometry in smgpu_iter_begin once the look-ahead has done every tile: the
    end-of-iteration reduction parked for that launch must still close ITS iteration (the per-iteration residual /
    nFrozenPoints records used to slip by one iteration).  world = 1 through the multi-rank path against the serial loop."""
    import socket
    import torch.distributed as dist
    from smoothmesh_amd import SmoothEngine, default_params
    from smoothmesh_amd.halo import DistributedSmoother
    from smoothmesh_amd.meshgen import hex_block, hex_subdomain
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        sub = hex_subdomain((14, 12, 10), (1, 1, 1), 0, jitter=0.3, seed=5)
        ds = DistributedSmoother(sub, device=0, overlap=overlap)
        prm = default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False)
        ds.set_params(prm)
        n_d, res_d, frz_d = ds.iterate(9, 0.0)
        eng = SmoothEngine(hex_block(14, 12, 10, lengths=(1.0, 1.0, 1.0), jitter=0.3, seed=5), device=0)
        eng.set_params(prm)
        n_s, res_s, frz_s = eng.iterate(9, 0.0)
        assert n_d == n_s == 9
        assert np.array_equal(frz_d, frz_s)
        assert np.array_equal(res_d, res_s)
        assert np.array_equal(ds.get_points(), eng.get_points())
    finally:
        dist.destroy_process_group()


def test_direct_rccl_exchange_of_the_python_driver():
    """grouped ncclSend / ncclRecv on the engine's stream (rccl_direct.py): communicator beside torch's, start-up self-check,
    same results as the serial loop (scripts/check_direct_exchange.py; one rank -- the box has one GPU)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_direct_exchange.py")], capture_output=True, text=True,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "direct exchange: ok" in r.stdout


@pytest.mark.parametrize("transport", ["rccl", "push"])
def test_every_arrangement_of_the_iteration_gives_the_same_bits(transport):
    """one rank of eight with its real halo tables and a self-exchange (scripts/check_arrangements.py): one kernel per step in
    order / with an exchange stream, the multi-role launches (k_geom_halo / k_smooth_halo on tiles of the shared points) in order,
    with an exchange stream ordered around whole launches, and flagged (the exchanges next to the launches, ordered by flag words)
    -- bit-identical coordinates, residuals and nFrozenPoints; the same with the peer-store transport onto the own receive slots"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if transport == "push":
        env["SMOOTHMESH_EXCHANGE"] = "push"
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_arrangements.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "arrangements: ok" in r.stdout and "DIFFERENT" not in r.stdout
    assert r.stdout.count("same bits") == (1 if transport == "push" else 4)


@pytest.mark.parametrize("world", [2, 8])
def test_distributed_smoother_polyhedral_over_rccl(world):
    """The production transport: one process per GPU, RCCL (send / recv groups on the engines' streams, torch's collectives
    for the set-up), decomposed polyhedral mesh against the oracle's MultiDomain.  Needs `world` GPUs: skipped on the 1-GPU
    development box (where RCCL refuses two ranks on one device) -- it is here for any node that has them."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SMOOTHMESH_SHARE_GPU", None); env.pop("SMOOTHMESH_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", "29527", os.path.join(root, "scripts", "check_dist_poly.py")],
                       capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ok ") == 4 * world and "BAD" not in r.stdout


def _graded_case(oracle_lib, grid, variant, constraints=False):
    """the exactly graded block of tests/test_sync_tie_rule.py (dx = dy / 2, plane points moved by binary fractions: every
    comparison of the +-x neighbours of a processor-plane point is an exact tie) cut into `grid` boxes"""
    from test_sync_tie_rule import graded_block
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import decompose, grid_partition, shared_point_table
    mesh = graded_block(32, 16, 16)
    world = grid[0] * grid[1] * grid[2]
    subs = decompose(mesh, grid_partition(mesh, grid), world)
    ser = oracle_lib.Oracle(mesh)
    prm = default_params(ser.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    ser.set_params(prm)
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    for o in orcs:
        o.set_params(prm)
    mo = oracle_lib.MultiOracle(orcs, *shared_point_table(subs))
    mo.set_sync_variant(variant)
    return mesh, subs, ser, orcs, mo, prm


@pytest.mark.parametrize("grid,overlap,inline", [((2, 1, 1), 0, 0), ((2, 2, 2), 0, 0), ((2, 2, 2), 1, 0), ((2, 1, 1), 0, 1), ((2, 2, 1), 0, 1)])
def test_tie_rule_master_fold_equals_serial_on_a_graded_block(oracle_lib, monkeypatch, grid, overlap, inline):
    """SM.C:388-478 with syncTools::syncPointList's master fold (smgpu_set_sync_variant, default): on a mesh where the two
    closest neighbours of every processor-plane point are at bit-equal distance the decomposed run equals the SERIAL oracle at
    every point (<= 1e-13: the cell-centre sums are formed per rank), equals the oracle's MultiDomain bit for bit, and the
    copies of a shared point stay identical.  Two-sharer kernel (k_halo_combineA2), 16-lane multi-sharer form (2x2x2: edge and
    centre points with 4 / 8 sharers), also with the freeze flags packed inside the smoothing kernel."""
    from smoothmesh_amd.halo import LocalMultiSmoother
    if inline:
        monkeypatch.setenv("SMGPU_HALO_INLINE_PACKF", "1")
    mesh, subs, ser, orcs, mo, prm = _graded_case(oracle_lib, grid, "master")
    ms = LocalMultiSmoother(subs, device=0, overlap=bool(overlap))
    ms.set_params(prm)
    iters = 3
    ser.iterate(iters, 0.0)
    mo.iterate(iters, 0.0)
    ms.iterate(iters, 0.0)
    sp = ser.points().reshape(-1, 3)
    for s, o, pts in zip(subs, orcs, ms.get_points()):
        assert np.array_equal(pts, o.points())                                        # the oracle's MultiDomain, bit for bit
        assert np.max(np.abs(pts.reshape(-1, 3) - sp[s.pointProcAddressing])) <= 1e-13     # ... and the serial run
    g = np.concatenate([s.pointProcAddressing for s in subs])
    allp = np.concatenate([p.reshape(-1, 3) for p in ms.get_points()])
    order = np.argsort(g, kind="stable")
    same = g[order][1:] == g[order][:-1]
    assert same.any() and np.array_equal(allp[order][1:][same], allp[order][:-1][same])


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)])
def test_tie_rule_own_fold_switch_matches_its_oracle_and_not_the_serial_run(oracle_lib, grid):
    """the A/B switch (SMGPU_SYNC_OWN): every sharer folds onto its own value, as rounds 1-3 did -- equal to the oracle's
    MultiDomain with the same switch, and visibly NOT the serial result on the processor plane"""
    from smoothmesh_amd.halo import LocalMultiSmoother
    mesh, subs, ser, orcs, mo, prm = _graded_case(oracle_lib, grid, "own")
    ms = LocalMultiSmoother(subs, device=0, overlap=False)
    ms.set_sync_variant("own")
    ms.set_params(prm)
    ser.iterate(1, 0.0); mo.iterate(1, 0.0); ms.iterate(1, 0.0)
    sp = ser.points().reshape(-1, 3)
    worst = 0.0
    for s, o, pts in zip(subs, orcs, ms.get_points()):
        assert np.array_equal(pts, o.points())
        worst = max(worst, float(np.max(np.abs(pts.reshape(-1, 3) - sp[s.pointProcAddressing]))))
    assert 1e-5 < worst < prm.maxStepLength


def test_tie_rule_with_layers_on_an_unjittered_block(oracle_lib):
    """the magnitude folds of the layer treatment (maxMagSqr of the propagated normals OBB.C:359-365 in the host-side set-up,
    minMagSqr of the outer neighbour coordinates OBB.C:490-496 in k_halo_combineAL) on a block without jitter, where the
    candidates tie exactly: GPU = the oracle's MultiDomain, bit for bit, copies identical"""
    from smoothmesh_amd import LayerParams, default_params, patch_arrays
    from smoothmesh_amd.decompose import decompose, grid_partition, shared_point_table
    from smoothmesh_amd.halo import LocalMultiSmoother
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(12, 12, 12, jitter=0.0)
    pts = mesh.points.reshape(-1, 3)
    pts[6 + 13 * 6 + 169 * 6] += np.array([1 / 64, 1 / 128, 0.0])
    pts[6 + 13 * 3 + 169 * 2] += np.array([0.0, 1 / 64, 1 / 64])
    subs = decompose(mesh, grid_partition(mesh, (2, 2, 2)), 8)
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=False, faceAngleConstraint=False)
    for o in orcs:
        o.set_params(prm)
    mo = oracle_lib.MultiOracle(orcs, *shared_point_table(subs))
    patches = ("xmin", "ymax", "zmin")
    assert mo.setup_layers([patch_arrays(s.mesh, patches) for s in subs], 0.3, prm.minEdgeLength, 1.2, 1, 4)
    ms = LocalMultiSmoother(subs, device=0, overlap=False)
    ms.set_params(prm)
    assert ms.set_layers(LayerParams(layerPatches=patches, layerExpansionRatio=1.2), prm.minEdgeLength)
    mo.iterate(4, 0.0); ms.iterate(4, 0.0)
    for o, p in zip(orcs, ms.get_points()):
        assert np.array_equal(p, o.points())
