"""End-to-end drop-in check of the `smoothMesh` front-end on the GPU box: case directory in,
<time>/polyMesh/points out, log lines as the reference prints them (SM.C:2396-2409)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import rel_linf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "smoothmesh_amd", "bin", "smoothMesh")
LINE = re.compile(r"Smoothing iteration=(\d+) nFrozenPoints=(\d+) residual=(\S+)")


def _run(args):
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("fmt", ["ascii", "binary"])
def test_serial_case(tmp_path, oracle_lib, fmt):
    from smoothmesh_amd import default_params
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_case
    m = hex_block(9, 8, 7, jitter=0.3, seed=4)
    write_case(str(tmp_path), m, binary=(fmt == "binary"), writeFormat=fmt)
    out = _run(["-case", str(tmp_path), "-centroidalIters", "12", "-relTol", "0", "-minAngle", "40", "-writeInterval", "5"])
    o = oracle_lib.Oracle(m)
    o.set_params(default_params(o.mesh_stats()[0], minAngle=40.0))
    n, res, frz = o.iterate(12, 0.0)
    lines = LINE.findall(out)
    assert [int(a) for a, _, _ in lines] == list(range(1, 13))
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    assert np.allclose([float(c) for _, _, c in lines], res, rtol=1e-5)          # %g prints 6 digits
    assert "Maximum centroidalIters reached, stopping." in out and "End" in out
    # writes at 5, 10 (writeInterval) and 12 (stop), SM.C:2416
    assert sorted(d for d in os.listdir(tmp_path) if d.isdigit()) == ["10", "12", "5"]
    got = read_polymesh(str(tmp_path / "constant" / "polyMesh"), pointsDir=str(tmp_path / "12" / "polyMesh")).points
    tol = 0.0 if fmt == "binary" else 2e-10           # ascii output carries 10 significant digits (SM.C:2425)
    assert rel_linf(got, o.points()) <= max(tol, 1e-13)
    # restart from latestTime continues the series
    out2 = _run(["-case", str(tmp_path), "-centroidalIters", "3", "-relTol", "0", "-minAngle", "40"])
    assert "Create mesh for time = 12" in out2 and os.path.isdir(tmp_path / "15")


def test_compressed_case(tmp_path, oracle_lib):
    """writeCompression on in controlDict: the mesh is read from *.gz and points are written as points.gz"""
    from smoothmesh_amd import default_params
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_case
    m = hex_block(7, 6, 5, jitter=0.3, seed=5)
    write_case(str(tmp_path), m, binary=True, writeFormat="binary", writeCompression=True)
    assert "points.gz" in os.listdir(tmp_path / "constant" / "polyMesh")
    _run(["-case", str(tmp_path), "-centroidalIters", "4", "-relTol", "0"])
    assert os.listdir(tmp_path / "4" / "polyMesh") == ["points.gz"]
    o = oracle_lib.Oracle(m)
    o.set_params(default_params(o.mesh_stats()[0]))
    o.iterate(4, 0.0)
    got = read_polymesh(str(tmp_path / "constant" / "polyMesh"), pointsDir=str(tmp_path / "4" / "polyMesh")).points
    assert rel_linf(got, o.points()) <= 1e-13


def test_layer_patches_case(tmp_path, oracle_lib):
    """-layerPatches & co. (SM.C:1749-1775): the boundary layer treatment through the command line"""
    from smoothmesh_amd import default_params, patch_arrays
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_case
    m = hex_block(10, 8, 7, jitter=0.25, seed=6)
    write_case(str(tmp_path), m, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-centroidalIters", "8", "-relTol", "0", "-layerPatches", '(xmin "y.*")',
                "-layerExpansionRatio", "1.2", "-maxLayers", "3", "-faceAngleConstraint", "false"])
    assert "Enabled boundary layer treatment" in out and "layerExpansionRatio      1.2" in out
    assert "WARNING: Boundary layer treatment will be done without boundary point smoothing" in out
    o = oracle_lib.Oracle(m)
    prm = default_params(o.mesh_stats()[0], faceAngleConstraint=False)
    o.set_params(prm)
    st, sz, kd, sel = patch_arrays(m, ["xmin", '"y.*"'])
    assert sel.tolist() == [1, 0, 1, 1, 0, 0]
    assert o.setup_layers(st, sz, kd, sel, 0.3, prm.minEdgeLength, 1.2, 1, 3)
    n, res, frz = o.iterate(8, 0.0)
    lines = LINE.findall(out)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    got = read_polymesh(str(tmp_path / "constant" / "polyMesh"), pointsDir=str(tmp_path / "8" / "polyMesh")).points
    assert rel_linf(got, o.points()) <= 1e-13
    # no such patch -> treatment disabled, as in the reference (SM.C:2025-2033)
    out2 = _run(["-case", str(tmp_path), "-time", "constant", "-centroidalIters", "1", "-relTol", "0", "-layerPatches", "nosuch"])
    assert "Patches for boundary layer treatment: none" in out2 and "Boundary layer treatment is disabled" in out2


def test_boundary_point_smoothing_case(tmp_path, oracle_lib):
    """constant/geometry/{targetSurfaces,initEdges,targetEdges}.obj present -> boundary points are projected to the target
    (SM.C:2080-2093, 2307-2357); the classification lists are written with the mesh and read back on a restart"""
    from bnd_cases import boundary_inputs, make_pair, scale_about_centre, tangential_jitter
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_case
    from smoothmesh_amd.surfgen import read_obj_edges, read_obj_surface, write_obj_edges, write_obj_surface
    m = tangential_jitter(hex_block(8, jitter=0.2, seed=3), 0.02, seed=4)
    write_case(str(tmp_path), m, binary=True, writeFormat="binary")
    init, target, surf = boundary_inputs(8, 3, warp=scale_about_centre(1.02))
    geo = tmp_path / "constant" / "geometry"
    os.makedirs(geo)
    write_obj_edges(str(geo / "initEdges.obj"), *init)
    write_obj_edges(str(geo / "targetEdges.obj"), *target)
    # quads in the file: the reader triangulates them as fans
    with open(geo / "targetSurfaces.obj", "w") as f:
        for p in surf[0]:
            f.write("v %.17g %.17g %.17g\n" % tuple(p))
        for k in range(0, len(surf[1]), 2):
            a, b = surf[1][k], surf[1][k + 1]                   # (q0 q1 q2), (q0 q2 q3)
            f.write("f %d %d %d %d\n" % (a[0] + 1, a[1] + 1, a[2] + 1, b[2] + 1))
    out = _run(["-case", str(tmp_path), "-centroidalIters", "10", "-relTol", "0", "-internalSmoothingBlendingFraction", "0.3"])
    assert "Enabled boundary point smoothing" in out
    assert "- Detected number of corner points: 8" in out and "- Detected number of feature edge points: 84" in out
    assert "Detected number of target edge mesh strings: 12" in out
    # what the front-end read is what the Python readers read
    init_r, target_r, surf_r = read_obj_edges(str(geo / "initEdges.obj")), read_obj_edges(str(geo / "targetEdges.obj")), read_obj_surface(str(geo / "targetSurfaces.obj"))
    assert np.array_equal(surf_r[1], surf[1]) and np.array_equal(init_r[1], init[1])
    o = make_pair(m, oracle_lib, init_r, target_r, surf_r, constraints=True, engine=False, blend=0.3)[0]
    n, res, frz = o.iterate(10, 0.0)
    lines = LINE.findall(out)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    got = read_polymesh(str(tmp_path / "constant" / "polyMesh"), pointsDir=str(tmp_path / "10" / "polyMesh")).points
    assert rel_linf(got, o.points()) <= 1e-13
    assert os.path.exists(tmp_path / "10" / "isCornerPoint") and os.path.exists(tmp_path / "10" / "isFeatureEdgePoint")
    # restart: the lists are found and used (the points have left the initial edges by now)
    out2 = _run(["-case", str(tmp_path), "-centroidalIters", "2", "-relTol", "0", "-internalSmoothingBlendingFraction", "0.3"])
    assert "Found corners and feature edges in isCornerPoint and isFeatureEdgePoint files" in out2
    assert "- Detected number of corner points: 8" in out2 and "- Detected number of feature edge points: 84" in out2


def test_parallel_boundary_point_smoothing_case(tmp_path, oracle_lib):
    """mpirun ... smoothMesh -parallel with constant/geometry/*.obj (the reference's run_parallel of testcase2-8): every
    sub-domain projects its boundary points, the shared ones from synchronised inputs; per-processor results equal the oracle's
    MultiDomain; the classification lists are written per processor"""
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface, write_obj_edges, write_obj_surface
    from test_oracle_boundary import _multi_boundary_case
    grid = (2, 2, 1)
    mo, orcs, subs, _, hi = _multi_boundary_case(oracle_lib, grid, (4, 5, 4), 0.25, True, blend=0.3)
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    geo = tmp_path / "constant" / "geometry"
    os.makedirs(geo)
    write_obj_edges(str(geo / "initEdges.obj"), *box_feature_edges(8, hi=hi))
    write_obj_surface(str(geo / "targetSurfaces.obj"), *box_surface(4, hi=hi))
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "6", "-relTol", "0", "-internalSmoothingBlendingFraction", "0.3"])
    assert "Enabled boundary point smoothing" in out
    n, res, frz = mo.iterate(6, 0.0)
    lines = LINE.findall(out)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    for s_, o in zip(subs, orcs):
        d = tmp_path / f"processor{s_.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "6" / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13
        assert os.path.exists(d / "6" / "isCornerPoint") and os.path.exists(d / "6" / "isFeatureEdgePoint")
    # boundary point smoothing stays off when no patch is to be smoothed (SM.C:2080-2093)
    out3 = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "1", "-smoothingPatches", "()"])
    assert "Boundary point smoothing is disabled" in out3


def test_relTol_stop_and_option_errors(tmp_path):
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import write_case
    write_case(str(tmp_path), hex_block(4))
    out = _run(["-case", str(tmp_path)])
    assert "Smoothing iteration=1 nFrozenPoints=98 residual=0" in out and "Residual reached relTol, stopping." in out
    assert os.path.isdir(tmp_path / "1")
    r = subprocess.run([BIN, "-case", str(tmp_path), "-noSuchOption", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "Wrong option" in r.stdout
    # boundary point smoothing needs the target surface AND the initial edges (SM.C:2080-2093): one file alone leaves it off
    os.makedirs(tmp_path / "constant" / "geometry")
    (tmp_path / "constant" / "geometry" / "targetSurfaces.obj").write_text("# empty\n")
    out = _run(["-case", str(tmp_path)])
    assert "Boundary point smoothing is disabled. Missing smoothingPatches, or one or both of files:" in out
    # both present but unusable (no triangles): an error, not a silent skip
    (tmp_path / "constant" / "geometry" / "initEdges.obj").write_text("v 0 0 0\nv 1 0 0\nl 1 2\n")
    r = subprocess.run([BIN, "-case", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "did not enable" in r.stdout
    (tmp_path / "constant" / "geometry" / "targetSurfaces.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 7\n")
    r = subprocess.run([BIN, "-case", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "vertex reference out of range" in r.stdout


def test_parallel_case(tmp_path, oracle_lib):
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    grid = (2, 2, 1)
    subs = [hex_subdomain((5, 4, 6), grid, r, jitter=0.3, seed=6) for r in range(4)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "7", "-relTol", "0"])
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs))
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    n, res, frz = mo.iterate(7, 0.0)
    lines = LINE.findall(out)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "7" / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13


def test_parallel_case_with_layer_patches(tmp_path, oracle_lib):
    """the reference's own parallel test (testcase/run_parallel:18): mpirun ... -parallel ... -layerPatches"""
    from smoothmesh_amd import default_params, patch_arrays
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    grid = (2, 2, 1)
    subs = [hex_subdomain((6, 5, 4), grid, r, jitter=0.25, seed=7) for r in range(4)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "6", "-relTol", "0", "-layerPatches", '(xmin "y.*")',
                "-layerExpansionRatio", "1.2", "-faceAngleConstraint", "false"])
    assert "Enabled boundary layer treatment" in out
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), faceAngleConstraint=False)
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    assert mo.setup_layers([patch_arrays(s.mesh, ["xmin", '"y.*"']) for s in subs], 0.3, prm.minEdgeLength, 1.2, 1, 4)
    n, res, frz = mo.iterate(6, 0.0)
    lines = LINE.findall(out)
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "6" / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13


@pytest.mark.parametrize("relTol", ["0", "1.5"])
def test_parallel_polyhedral_case_one_process_per_rank(tmp_path, oracle_lib, relTol):
    """`smoothMesh -parallel` = one process per processorN/ (the reference's `mpirun -np 4 smoothMesh -parallel`,
    testcase/run_parallel:19) on a decomposed POLYHEDRAL case; on this one-GPU box the ranks share the device and the records
    travel through the debug transport (RCCL needs a device per rank).  relTol 0: no host synchronisation inside a chunk, the
    per-iteration values are gathered afterwards; relTol > 0: reduced every iteration, the loop stops like the reference's."""
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.polymesh import cavity_subdomain, read_polymesh, write_decomposed_case
    grid = (2, 2, 1)
    subs = [cavity_subdomain(12, grid, r, jitter=0.2, seed=4) for r in range(4)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "9", "-relTol", relTol, "-writeInterval", "4"])
    assert "nProcs : 4" in out and "debug transport" in out
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs))
    for o in orcs:
        o.set_params(prm)
    mo = oracle_lib.MultiOracle(orcs, *shared_point_table(subs))
    n, res, frz = mo.iterate(9, float(relTol))
    lines = LINE.findall(out)
    assert len(lines) == n and (n < 9) == (relTol != "0")
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    assert np.allclose([float(c) for _, _, c in lines], res, rtol=1e-5)
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / str(n) / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13


@pytest.mark.parametrize("constraints", [False, True])
def test_parallel_case_over_peer_stores(tmp_path, oracle_lib, monkeypatch, constraints):
    """SMOOTHMESH_TRANSPORT=push: the ranks of `smoothMesh -parallel` map each other's receive buffers (hipIpc) and the pack
    kernels store the records there themselves (include/smgpu.h, smgpu_push_desc); the front-end moves nothing between the
    smgpu_iter_* calls.  Four ranks on this box's one device (IPC between processes works on one device too), a polyhedral
    case, against the oracle's MultiDomain."""
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.polymesh import cavity_subdomain, read_polymesh, write_decomposed_case
    monkeypatch.setenv("SMOOTHMESH_TRANSPORT", "push")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    grid = (2, 2, 1)
    subs = [cavity_subdomain(10, grid, r, jitter=0.2, seed=9) for r in range(4)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    flag = "true" if constraints else "false"
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "7", "-relTol", "0", "-edgeAngleConstraint", flag,
                "-faceAngleConstraint", flag])
    assert "nProcs : 4" in out and "peer stores" in out and "debug transport" not in out
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    for o in orcs:
        o.set_params(prm)
    mo = oracle_lib.MultiOracle(orcs, *shared_point_table(subs))
    n, res, frz = mo.iterate(7, 0.0)
    lines = LINE.findall(out)
    assert len(lines) == n == 7
    assert [int(b) for _, b, _ in lines] == frz.tolist()
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "7" / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13


def test_parallel_single_rank_initialises_rccl(tmp_path, oracle_lib):
    """one processor0/ directory: the rank is its own process and, having a device of its own, brings up RCCL
    (ncclCommInitRank with one rank); results equal the serial run's"""
    from smoothmesh_amd import default_params
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    subs = [hex_subdomain((7, 6, 5), (1, 1, 1), 0, jitter=0.3, seed=2)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "5", "-relTol", "0"])
    assert "nProcs : 1" in out and "debug transport" not in out
    o = oracle_lib.Oracle(subs[0].mesh)
    o.set_params(default_params(o.mesh_stats()[0]))
    n, res, frz = o.iterate(5, 0.0)
    assert [int(b) for _, b, _ in LINE.findall(out)] == frz.tolist()
    got = read_polymesh(str(tmp_path / "processor0" / "constant" / "polyMesh"), pointsDir=str(tmp_path / "processor0" / "5" / "polyMesh")).points
    assert rel_linf(got, o.points()) <= 1e-13


def test_parallel_case_over_rccl_when_the_box_has_two_gpus(tmp_path, oracle_lib, monkeypatch):
    """two ranks, two devices, SMOOTHMESH_TRANSPORT=rccl: the grouped ncclSend / ncclRecv exchange between DIFFERENT ranks (and
    its one-shot self-check against the slot layout) -- the path a 1-GPU box can never reach.  Skipped there."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    from smoothmesh_amd import default_params
    from smoothmesh_amd.decompose import shared_point_table
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.polymesh import read_polymesh, write_decomposed_case
    monkeypatch.setenv("SMOOTHMESH_TRANSPORT", "rccl")
    subs = [hex_subdomain((6, 5, 4), (2, 1, 1), r, jitter=0.3, seed=9) for r in range(2)]
    write_decomposed_case(str(tmp_path), subs, binary=True, writeFormat="binary")
    out = _run(["-case", str(tmp_path), "-parallel", "-centroidalIters", "6", "-relTol", "0"])
    assert "debug transport" not in out and "self-check failed" not in out
    orcs = [oracle_lib.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs))
    for o in orcs:
        o.set_params(prm)
    off, dom, loc = shared_point_table(subs)
    mo = oracle_lib.MultiOracle(orcs, off, dom, loc)
    n, res, frz = mo.iterate(6, 0.0)
    assert [int(b) for _, b, _ in LINE.findall(out)] == frz.tolist()
    for s, o in zip(subs, orcs):
        d = tmp_path / f"processor{s.rank}"
        got = read_polymesh(str(d / "constant" / "polyMesh"), pointsDir=str(d / "6" / "polyMesh")).points
        assert rel_linf(got, o.points()) <= 1e-13
