#!/usr/bin/env python3
"""Writes tests/golden/hex6_boundary_seed5.npz: inputs + ORACLE outputs of the boundary point smoothing (points after
1/5/15 iterations, per-iteration nFrozenPoints and residual, classification) for a 6^3 jittered hex block whose boundary
is smoothed onto its surface scaled by 1.03 (target surface: 3x3 quads per side; feature edges: 6 segments per block
edge), defaults otherwise (constraints on), internalSmoothingBlendingFraction 0.4.  Regression vectors for the oracle
and the HIP path; NOT outputs of the real reference (it needs OpenFOAM, absent here -- parity unpinned)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from bnd_cases import boundary_inputs, make_pair, scale_about_centre, tangential_jitter  # noqa: E402
from oracle import oracle_ffi  # noqa: E402
from smoothmesh_amd.meshgen import hex_block  # noqa: E402


def case():
    m = tangential_jitter(hex_block(6, jitter=0.3, seed=5), 0.03, seed=6)
    init, target, surf = boundary_inputs(6, 3, warp=scale_about_centre(1.03))
    return m, init, target, surf


if __name__ == "__main__":
    m, init, target, surf = case()
    out = {"points0": np.array(m.points).copy()}
    o = make_pair(m, oracle_ffi, init, target, surf, constraints=True, engine=False, blend=0.4)[0]
    f = o.boundary_fields()
    out["isCornerPoint"], out["isFeatureEdgePoint"] = f["isCornerPoint"], f["isFeatureEdgePoint"]
    out["pointStrings"], out["innerMap"] = f["pointStrings"], f["innerMap"]
    res_all, frz_all = [], []
    for tag, iters in (("1", 1), ("5", 4), ("15", 10)):
        n, res, frz = o.iterate(iters, 0.0)
        res_all.append(res); frz_all.append(frz)
        out["points" + tag] = o.points()
    out["residual"] = np.concatenate(res_all)
    out["nFrozen"] = np.concatenate(frz_all)
    np.savez_compressed(os.path.join(HERE, "hex6_boundary_seed5.npz"), **out)
    print("written", {k: v.shape for k, v in out.items()})
