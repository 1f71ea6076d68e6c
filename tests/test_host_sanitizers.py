"""The product's host-only C++ (addressing build, tile tables, layer and boundary smoothing set-up incl. the triangle hierarchy, polyMesh I/O, mesh generator) under
AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "smoothmesh_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    src = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + \
          [os.path.join(CSRC, f) for f in ("topology.cpp", "tiles.cpp", "layers.cpp", "boundary.cpp", "host/meshgen.cpp", "host/polymesh_io.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", exe] + src + ["-lz"])
    # polyMesh I/O with the lists cut into pieces of a few records for five threads (round trips + truncated files), and serially
    for k, io in enumerate(({"SMHOST_IO_THREADS": "5", "SMHOST_IO_GRAIN": "64"}, {"SMHOST_IO_THREADS": "1"})):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1", **io)
        r = subprocess.run([exe, "14", str(tmp_path / f"case{k}")], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.startswith("ok points") and "refused 48" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_polymesh_io_is_race_free(tmp_path):
    """the same driver under ThreadSanitizer: per-thread pieces of the ascii readers, the pwrite writers, the parallel copies"""
    exe = str(tmp_path / "host_tsan")
    src = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + \
          [os.path.join(CSRC, f) for f in ("topology.cpp", "tiles.cpp", "layers.cpp", "boundary.cpp", "host/meshgen.cpp", "host/polymesh_io.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-pthread", "-fsanitize=thread", "-o", exe] + src + ["-lz"])
    env = dict(os.environ, SMHOST_IO_THREADS="6", SMHOST_IO_GRAIN="64", SMGPU_HOST_THREADS="3", SMGPU_HOST_GRAIN="64")
    r = subprocess.run([exe, "10", str(tmp_path / "case")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert r.stdout.startswith("ok points") and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def _dump_mesh(path, n):
    import numpy as np
    from smoothmesh_amd.polymesh import cavity_mesh
    m = cavity_mesh(n, jitter=0.2, seed=4)
    with open(path, "wb") as f:
        np.array([m.nPoints, m.nCells, m.nFaces, m.nInternalFaces], dtype=np.int32).tofile(f)
        for a, t in ((m.points, np.float64), (m.faceOffsets, np.int32), (m.facePoints, np.int32), (m.owner, np.int32), (m.neighbour, np.int32),
                     (m.find_internal_points(), np.uint8)):
            np.ascontiguousarray(a, dtype=t).tofile(f)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_table_builds_equal_the_serial_ones_and_are_race_free(tmp_path):
    """smgpu_create builds the addressing and the tile tables on host threads (csrc/parallel.hpp): index ranges in order, per-range
    results concatenated.  With several ranges forced on a small mesh (SMGPU_HOST_GRAIN) every table must come out byte for byte
    as the serial build makes it (scripts/native/setup_bench prints a checksum per table), and ThreadSanitizer must stay quiet."""
    mesh = str(tmp_path / "mesh.bin")
    _dump_mesh(mesh, 20)
    src = [os.path.join(ROOT, "scripts", "native", "setup_bench.cpp"), os.path.join(CSRC, "topology.cpp"), os.path.join(CSRC, "tiles.cpp")]
    exe, tsan = str(tmp_path / "setup_bench"), str(tmp_path / "setup_bench_tsan")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", CSRC, "-o", exe] + src)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", CSRC, "-o", tsan] + src)
    # the reference output: one host thread, the tile tables built AFTER the addressing
    serial = subprocess.run([exe, mesh], capture_output=True, text=True, env=dict(os.environ, SMGPU_HOST_THREADS="1", SETUP_BENCH_PIPELINED="0"), timeout=600)
    assert serial.returncode == 0 and "cellFacesGeom.val" in serial.stdout
    # everything below runs as smgpu_create does: the geometry / smoothing tile tables started from Topology::build's afterCells /
    # afterPoints hooks while build() is still filling the remaining addressing
    piped = subprocess.run([exe, mesh], capture_output=True, text=True, env=dict(os.environ, SMGPU_HOST_THREADS="1"), timeout=600)
    assert piped.returncode == 0 and piped.stdout == serial.stdout and "started from its hooks" in piped.stderr
    for threads, grain in (("3", "64"), ("8", "17")):
        env = dict(os.environ, SMGPU_HOST_THREADS=threads, SMGPU_HOST_GRAIN=grain)
        r = subprocess.run([exe, mesh], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and r.stdout == serial.stdout, (threads, grain)
    r = subprocess.run([tsan, mesh], capture_output=True, text=True, env=dict(os.environ, SMGPU_HOST_THREADS="4", SMGPU_HOST_GRAIN="64"), timeout=900)
    assert r.returncode == 0 and r.stdout == serial.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
