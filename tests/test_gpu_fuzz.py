"""Randomised GPU-vs-oracle parity sweep (scripts/fuzz_parity.py): random meshes (hex, polyhedral, decomposed), jitter up
to near-inversion, random parameters, constraints and layer patches; every case must be bit-compatible."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "10", str(seed)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 10


@pytest.mark.parametrize("kinds,seed", [("irregular", 51), ("tied", 52), ("irregular,tied", 53)])
def test_random_irregular_and_tied_cases(kinds, seed):
    """the sweep restricted to irregular decompositions (breadth-first grown / random cellRank maps, 2..8 ranks, disconnected
    pieces, hex and polyhedral) and to unjittered / exactly graded blocks on binary fractions (exact ties in the closest-point syncs)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "10", str(seed)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, FUZZ_KINDS=kinds))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 10


@pytest.mark.parametrize("seed", [61, 62])
def test_random_cases_in_other_units_and_far_from_the_origin(seed):
    """the sweep on meshes scaled by 1e-6 .. 1e6 and moved up to 1e7 mesh sizes away from the origin, constraints on: the f32
    filters see f64 differences only and must stay on the safe side (bit-equal decisions) whatever the coordinates' magnitude"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "12", str(seed)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, FUZZ_KINDS="affine"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 12


@pytest.mark.parametrize("env", [{"SMGPU_WALK": "fix", "SMGPU_WALK_PACK": "0"}, {"SMGPU_WALK": "fix"}, {"SMGPU_WALK": "fix", "SMGPU_WALK_STAR": "0", "SMGPU_WALK_BLOCKS": "7"},
                                 {"SMGPU_WALK": "host"}, {"SMGPU_FA_LISTS": "0", "SMGPU_FILTER": "0"}, {"SMGPU_WALK": "fix", "SMGPU_FA_SIDE_EXACT": "0"}])
def test_random_cases_under_walk_knobs(env):
    """the same sweep with the face-angle walk forced to the fixed-point device replay (with the star form of the predicates --
    one job per step -- and with the default packed form; also with the gather-form predicates and
    an odd number of workgroups in the persistent launch), to the host replay, and without filters / lists"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "8", "31"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 8
