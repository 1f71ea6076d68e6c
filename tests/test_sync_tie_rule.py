"""The multi-rank tie rule of findClosestPoints (SM.C:388-478, isCloserPoint :246-272) on meshes where ties DO occur.

syncTools::syncPointList of the OpenFOAM versions the reference builds against (globalMeshData::syncData) folds a shared
point's values once, on the master (lowest processor), in ascending processor order, and hands every sharer the SAME
value.  On an exactly graded block cut by a processor plane both +-x neighbours of a plane point are at bit-equal
distance; only with that fold does a rank receive the other rank's equal-length vector, isCloserPoint's "same distance,
different coordinates" case fires, and the 2-rank run reproduces the serial aspect-ratio blend.  These tests pin that
(CPU: the oracle's MultiDomain and the product's host-side tables through the oracle rank engine)."""
import numpy as np
import pytest

from conftest import rel_linf


def graded_block(nx=16, ny=8, nz=8, shift=True):
    """unit cube, dx = dy / 2 = dz / 2; the internal points of the plane x = 0.5 moved in y by exact binary fractions (their +-x
    neighbours stay, so the two distances stay bit-equal)"""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(nx, ny, nz, jitter=0.0)
    if shift:
        pts = mesh.points.reshape(-1, 3)
        i = nx // 2
        for k in range(1, nz):
            for j in range(1, ny):
                p = i + j * (nx + 1) + k * (nx + 1) * (ny + 1)
                pts[p, 1] += (((3 * j + 5 * k) % 7) - 3) / 256.0
    return mesh


def _serial(oracle_lib, mesh, iters, constraints=False):
    from smoothmesh_amd import default_params
    o = oracle_lib.Oracle(mesh)
    prm = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(prm)
    o.iterate(iters, 0.0)
    return o.points(), prm


def _multi(oracle_lib, mesh, grid, prm, iters, variant):
    from smoothmesh_amd.decompose import decompose, grid_partition, shared_point_table
    world = grid[0] * grid[1] * grid[2]
    subs = decompose(mesh, grid_partition(mesh, grid), world)
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    for o in orcs:
        o.set_params(prm)
    mo = oracle_lib.MultiOracle(orcs, *shared_point_table(subs))
    mo.set_sync_variant(variant)
    mo.iterate(iters, 0.0)
    out = np.full_like(mesh.points.reshape(-1, 3), np.nan)
    dup_equal = True
    for s, o in zip(subs, orcs):
        p = o.points().reshape(-1, 3)
        g = s.pointProcAddressing
        seen = ~np.isnan(out[g, 0])
        dup_equal &= bool(np.array_equal(out[g][seen], p[seen]))
        out[g] = p
    return out, dup_equal, subs


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)])
def test_master_fold_reproduces_the_serial_blend_on_a_graded_block(oracle_lib, grid):
    mesh = graded_block()
    ser, prm = _serial(oracle_lib, mesh, 1)
    ser = ser.reshape(-1, 3)
    par, dup_equal, _ = _multi(oracle_lib, mesh, grid, prm, 1, "master")
    assert dup_equal                                       # every sharer ends with the same coordinates
    assert np.max(np.abs(par - ser)) <= 1e-13              # (partial sums in rank order instead of cell order: 1 ulp)
    # ... and visibly not under the own-value fold: neither rank receives the other's equal-length vector, hasCommonCell
    # stays true and the aspect-ratio blend of the plane points is dropped
    own, _, _ = _multi(oracle_lib, mesh, grid, prm, 1, "own")
    d = np.abs(own - ser).max(axis=1)
    plane = np.isclose(mesh.points.reshape(-1, 3)[:, 0], 0.5) & (d > 1e-6)
    assert plane.sum() >= 30 and d.max() > 1e-4 and d.max() < prm.maxStepLength
    off_plane = ~np.isclose(mesh.points.reshape(-1, 3)[:, 0], 0.5)
    assert d[off_plane].max() <= 1e-13


@pytest.mark.parametrize("grid,iters", [((2, 1, 1), 5), ((1, 2, 2), 5), ((2, 2, 2), 3)])
def test_unjittered_uniform_block_ties_everywhere(oracle_lib, grid, iters):
    """a uniform block with a few displaced points: every edge-length comparison at a processor plane is an exact tie, over
    several iterations (the ties persist where the displacement has not arrived yet)"""
    from smoothmesh_amd.meshgen import hex_block
    mesh = hex_block(8, 8, 8, jitter=0.0)
    pts = mesh.points.reshape(-1, 3)
    for (i, j, k), dlt in {(4, 4, 4): (1 / 64, 0, 1 / 128), (4, 2, 5): (0, 1 / 64, 0), (3, 4, 4): (1 / 128, 1 / 128, 0)}.items():
        pts[i + 9 * j + 81 * k] += np.array(dlt)
    ser, prm = _serial(oracle_lib, mesh, iters)
    par, dup_equal, _ = _multi(oracle_lib, mesh, grid, prm, iters, "master")
    assert dup_equal
    assert np.max(np.abs(par - ser.reshape(-1, 3))) <= 1e-13


def test_isCloserPoint_receives_an_equal_length_vector(oracle_lib):
    """SM.C:242-244 'distance is equal but point coordinates are different': the case the three sequential syncs exist for"""
    a, b = np.array([-0.0625, 0.0, 0.0]), np.array([0.0625, 0.0, 0.0])
    assert oracle_lib.isCloserPoint(a, b) and oracle_lib.isCloserPoint(b, a) and not oracle_lib.isCloserPoint(a, a)


def test_rank_engine_tables_follow_the_master_fold(oracle_lib, tmp_path):
    """the product's host logic (HaloTables: sharers ascending by rank, -1 = this rank) drives the same fold through the oracle
    rank engine: world_size-2 gloo run of the graded block = the serial result"""
    import torch.multiprocessing as mp
    from test_halo_gloo import _free_port
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mesh = graded_block()
    ser, prm = _serial(oracle_lib, mesh, 2)
    ser = ser.reshape(-1, 3)
    for r in range(2):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert np.max(np.abs(d["pts"].reshape(-1, 3) - ser[d["gid"]])) <= 1e-13


def _worker(rank, world, port, out_dir):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle.oracle_ffi import Oracle, OracleRankEngine
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import decompose, grid_partition
    from smoothmesh_amd.halo import DistributedSmoother
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh = graded_block()
    sub = decompose(mesh, grid_partition(mesh, (2, 1, 1)), 2)[rank]
    ds = DistributedSmoother(sub, engine_factory=OracleRankEngine, torch_device=torch.device("cpu"))
    prm = default_params(Oracle(mesh).mesh_stats()[0], edgeAngleConstraint=False, faceAngleConstraint=False)
    ds.set_params(prm)
    ds.iterate(2, 0.0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pts=ds.get_points(), gid=sub.pointProcAddressing)
    dist.barrier()
    dist.destroy_process_group()
