"""GPU part of the irregular-decomposition coverage (tests/test_irregular_partitions.py holds the cases and the CPU part):
scotch-like ragged cuts, a disconnected sub-domain, a rank inside another, a rank without any shared point, points with 3..8
sharers off the lattice pattern -- through every multi-rank host of the product: N engines on one device
(LocalMultiSmoother), one PROCESS per rank with the Python driver (DistributedSmoother), and `smoothMesh -parallel` on the
written processorN/ directories (the reference's `mpirun -np 3 smoothMesh -parallel`, testcase/run_parallel:19, on a
decomposePar-style case).  Expected = the oracle's MultiDomain with the same decomposition, bit for bit."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from test_irregular_partitions import CASES, _oracles, build_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "smoothmesh_amd", "bin", "smoothMesh")
LINE = re.compile(r"Smoothing iteration=(\d+) nFrozenPoints=(\d+) residual=(\S+)")


@pytest.mark.parametrize("constraints", [False, True])
@pytest.mark.parametrize("kind,nR,seed", CASES)
def test_local_multi_smoother_on_irregular_partitions(oracle_lib, kind, nR, seed, constraints):
    from smoothmesh_amd.halo import LocalMultiSmoother
    mesh, cr = build_case(kind, nR, seed)
    subs, ser, orcs, mo, prm = _oracles(oracle_lib, mesh, cr, nR, constraints)
    ms = LocalMultiSmoother(subs, device=0, overlap=False)
    assert ms.global_min_edge() == min(o.mesh_stats()[0] for o in orcs)
    ms.set_params(prm)
    n_o, res_o, frz_o = mo.iterate(6, 0.0)
    n_g, res_g, frz_g = ms.iterate(6, 0.0)
    assert n_o == n_g and np.array_equal(frz_o, frz_g)
    assert np.max(np.abs(res_o - res_g) / np.maximum(res_o, 1e-300)) <= 1e-10
    for o, pts in zip(orcs, ms.get_points()):
        assert np.array_equal(pts, o.points())
    # copies of a shared point are identical wherever the reference keeps them so (every sharer agrees on internal / boundary;
    # see test_irregular_partitions for the points where findInternalMeshPoints itself depends on the decomposition)
    gInt = mesh.find_internal_points().astype(bool)
    g = np.concatenate([s.pointProcAddressing for s in subs])
    agree = np.concatenate([s.mesh.find_internal_points().astype(bool) == gInt[s.pointProcAddressing] for s in subs])
    rogue = np.zeros(mesh.nPoints, bool)
    rogue[g[~agree]] = True
    if not rogue.any():
        allp = np.concatenate([p.reshape(-1, 3) for p in ms.get_points()])
        order = np.argsort(g, kind="stable")
        same = g[order][1:] == g[order][:-1]
        assert same.any() and np.array_equal(allp[order][1:][same], allp[order][:-1][same])


@pytest.mark.parametrize("kind,nR,seed", [("hex_island", 3, 31), ("poly_bfs", 5, 32), ("two_blocks", 3, 33), ("hex_baffle", 4, 34)])
@pytest.mark.parametrize("match", ["ids", "patches"])
def test_parallel_cli_on_irregular_processor_directories(tmp_path, oracle_lib, kind, nR, seed, match):
    """match = "patches": the sub-domains WITHOUT pointProcAddressing (a mesh made in parallel, e.g. by snappyHexMesh -parallel, has
    none): the front-end then finds the copies of a point through the processor patches themselves, vertex by vertex of the
    matching faces, as OpenFOAM's globalPoints does -- same tables, same result."""
    from smoothmesh_amd.decompose import decompose
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    mesh, cr = build_case(kind, nR, seed)
    subs, ser, orcs, mo, prm = _oracles(oracle_lib, mesh, cr, nR, True)
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    if match == "patches":
        for s in subs:
            os.remove(tmp_path / f"processor{s.rank}" / "constant" / "polyMesh" / "pointProcAddressing")
    r = subprocess.run([BIN, "-case", str(tmp_path), "-parallel", "-centroidalIters", "6", "-relTol", "0", "-minEdgeLength", repr(prm.minEdgeLength),
                        "-maxStepLength", repr(prm.maxStepLength)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"nProcs : {nR}" in r.stdout
    assert ("Shared points matched through the processor patches" in r.stdout) == (match == "patches")
    n, res, frz = mo.iterate(6, 0.0)
    lines = LINE.findall(r.stdout)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    assert np.allclose([float(c) for _, _, c in lines], res, rtol=1e-5)
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "6" / "polyMesh")).points
        assert np.array_equal(got.reshape(-1), o.points().reshape(-1))


@pytest.mark.parametrize("spec,world,port", [("two_blocks:41", 3, "29541"), ("poly_bfs:42", 3, "29543")])
def test_distributed_smoother_irregular_processes(spec, world, port):
    """one process per rank with the real engines (three ranks share this box's GPU; gloo carries the records): a rank without
    shared points must still take part in the collectives, ragged counts, in order and overlapped, constraints off and on"""
    env = dict(os.environ, SMOOTHMESH_SHARE_GPU="1", SMOOTHMESH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", CHECK_IRREGULAR=spec)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "scripts", "check_dist_poly.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ok ") == 4 * world and "BAD" not in r.stdout


@pytest.mark.parametrize("host", ["cli", "cli-no-ids", "engines"])
def test_baffle_between_ranks_with_layers_grown_from_it(tmp_path, oracle_lib, host):
    """The reference's testcase6 in parallel (`-layerPatches '(walls "baffle.*")'`, testcase6/run_parallel) with the wall BETWEEN
    ranks: the copies of a baffle point on its two sides are connected by no processor face, so syncTools::syncPointList does
    not combine them (OpenFOAM's globalPoints; decompose.shared_point_components) -- the layer normals summed over the sharers
    (OBB.C:184-190) stay one-sided there.  Through `smoothMesh -parallel` (its own C++ grouping) and through N engines on
    one device (halo.HaloTables), against the oracle's MultiDomain on the tables of decompose.shared_point_table."""
    from smoothmesh_amd import LayerParams, patch_arrays
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    mesh, cr = build_case("hex_baffle", 4, 35)
    subs, ser, orcs, mo, prm = _oracles(oracle_lib, mesh, cr, 4, True)
    pats = ['"baffle.*"', "ymax"]
    assert mo.setup_layers([patch_arrays(s.mesh, pats) for s in subs], 0.3, prm.minEdgeLength, 1.2, 1, 3)
    n, res, frz = mo.iterate(6, 0.0)
    if host.startswith("cli"):
        write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
        if host == "cli-no-ids":
            for s in subs:
                os.remove(tmp_path / f"processor{s.rank}" / "constant" / "polyMesh" / "pointProcAddressing")
        r = subprocess.run([BIN, "-case", str(tmp_path), "-parallel", "-centroidalIters", "6", "-relTol", "0", "-minEdgeLength", repr(prm.minEdgeLength),
                            "-maxStepLength", repr(prm.maxStepLength), "-layerPatches", '("baffle.*" ymax)', "-layerExpansionRatio", "1.2", "-maxLayers", "3"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "Enabled boundary layer treatment" in r.stdout
        assert [int(b) for _, b, _ in LINE.findall(r.stdout)] == frz.tolist()
        for s, o in zip(subs, orcs):
            d = tmp_path / f"processor{s.rank}"
            got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "6" / "polyMesh")).points
            assert np.array_equal(got.reshape(-1), o.points().reshape(-1))
    else:
        from smoothmesh_amd.halo import LocalMultiSmoother
        ms = LocalMultiSmoother(subs, device=0, overlap=False)
        ms.set_params(prm)
        assert ms.set_layers(LayerParams(layerPatches=tuple(pats), layerExpansionRatio=1.2, maxLayers=3), prm.minEdgeLength)
        n_g, res_g, frz_g = ms.iterate(6, 0.0)
        assert n_g == n and np.array_equal(frz_g, frz)
        for o, pts in zip(orcs, ms.get_points()):
            assert np.array_equal(pts, o.points())
