"""Product-side near-tie census (include/smgpu.h: smgpu_iter_stats::nNearTies, smgpu_get_near_ties): the engine counts the angle
comparisons (SM.C:923, 1367, 1391-1394, 1421-1424) whose two sides are 1 .. SMGPU_NEARTIE_ULPS (default 4) ulp apart -- the only
decisions the reference's acos (glibc; the engine's differs in the last bit for ~6 % of the arguments) could take the other way.
Checked against the oracle's census of the same comparisons (oracle run with the engine's acos, so both sides see the same bits)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
from smoothmesh_amd import SmoothEngine, default_params
from smoothmesh_amd.meshgen import hex_block
from oracle import oracle_ffi
mesh = hex_block(10, 10, 10, jitter=0.3, seed=9)          # tests/test_oracle_acos.py: thresholds at the block's own right angles
eng = SmoothEngine(mesh, device=0)
prm = default_params(eng.mesh_stats()[0], edgeAngleConstraint=True, faceAngleConstraint=True, minAngle=88.0, maxAngle=92.0)
eng.set_params(prm)
n, res, frz = eng.iterate(%(iters)d, 0.0)
per_iter = eng.last_near_ties.tolist()
tot = eng.near_ties()
prev = oracle_ffi.set_acos_variant("device")
o = oracle_ffi.Oracle(mesh)
o.set_params(prm)
oracle_ffi.acos_census_window(%(window)d)
oracle_ffi.acos_census(True)
no, reso, frzo = o.iterate(%(iters)d, 0.0)
cen = oracle_ffi.acos_census(False)
oracle_ffi.acos_census_window(4)
oracle_ffi.set_acos_variant(prev)
print(json.dumps({"per_iter": per_iter, "total": tot, "oracle": cen, "frz": frz.tolist(), "frzo": frzo.tolist(),
                  "same_points": bool(np.array_equal(eng.get_points(), o.points()))}))
"""


def _run(window, env, iters=5):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", PROG % {"root": ROOT, "window": window, "iters": iters}], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_no_near_ties_with_the_default_window():
    """the normal case: the census says no comparison of the run came within 4 ulp -- on the engine and on the oracle alike"""
    d = _run(4, {})
    assert d["same_points"] and d["frz"] == d["frzo"] and max(d["frz"]) > 0
    assert d["total"] == {"total": 0, "edge_angle": 0, "good_range": 0, "walk": 0} and d["per_iter"] == [0] * 5
    assert d["oracle"]["near"] == {"edge_angle": 0, "good_range": 0, "walk": 0} and d["oracle"]["within_8ulp"] == 0


@pytest.mark.parametrize("window", [2 ** 44, 2 ** 52])
def test_census_counts_what_the_oracle_counts(window):
    """a window wide enough to catch comparisons (2^44 ulp ~ 0.2 %% of an angle, 2^52: a factor of two), exact kernels on every
    element (SMGPU_FILTER=0: the f32 filters would decide most elements without the exact comparison): the engine's counts of the
    edge-angle test and of the good-range test ARE the oracle's; the walk's predicates are evaluated for every (active point,
    neighbour) pair up front where the reference's stack walk reaches only some of them, so that class is a superset"""
    d = _run(window, {"SMGPU_NEARTIE_ULPS": str(window), "SMGPU_FILTER": "0"})
    assert d["same_points"] and d["frz"] == d["frzo"]
    t, o = d["total"], d["oracle"]["near"]
    assert t["edge_angle"] == o["edge_angle"] > 0 and t["good_range"] == o["good_range"] > 0
    assert t["walk"] >= o["walk"] > 0
    assert sum(d["per_iter"]) == t["total"] == t["edge_angle"] + t["good_range"] + t["walk"]
