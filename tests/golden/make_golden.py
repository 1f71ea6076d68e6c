#!/usr/bin/env python3
"""Writes tests/golden/hex6_jitter03_seed7.npz: inputs + ORACLE outputs (points after 1/5/20
iterations, per-iteration nFrozenPoints and residual) for a 6^3 jittered hex block, defaults
(constraints on).  These are regression vectors for the oracle and the HIP path; they are NOT
outputs of the real reference (it needs OpenFOAM, absent here -- parity unpinned)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402
from smoothmesh_amd import default_params  # noqa: E402
from smoothmesh_amd.meshgen import hex_block  # noqa: E402

m = hex_block(6, jitter=0.3, seed=7)
out = {"points0": m.points.copy()}
o = oracle_ffi.Oracle(m)
o.set_params(default_params(o.mesh_stats()[0]))
res_all, frz_all = [], []
for tag, iters in (("1", 1), ("5", 4), ("20", 15)):
    n, res, frz = o.iterate(iters, 0.0)
    res_all.append(res); frz_all.append(frz)
    out["points" + tag] = o.points()
out["residual"] = np.concatenate(res_all)
out["nFrozen"] = np.concatenate(frz_all)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hex6_jitter03_seed7.npz"), **out)
print("written", {k: v.shape for k, v in out.items()})
