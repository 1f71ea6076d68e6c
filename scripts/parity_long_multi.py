#!/usr/bin/env python3
"""Long-run parity of the MULTI-RANK path: N engines on one device (halo.LocalMultiSmoother: the product's tables, pack / combine
kernels and exchange layout, the records staged on the device) against the oracle's MultiDomain over BASELINE's iteration counts,
compared every tenth of the run: identical nFrozenPoints series, coordinates bit-equal on every rank.  One JSON line per case:
    python scripts/parity_long_multi.py boxes:100 irregular:200
The oracle is test infrastructure (oracle/); nothing here is timed or shipped."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402
from smoothmesh_amd import default_params  # noqa: E402
from smoothmesh_amd.decompose import bfs_partition, decompose, shared_point_table  # noqa: E402
from smoothmesh_amd.halo import LocalMultiSmoother  # noqa: E402
from smoothmesh_amd.meshgen import hex_subdomain  # noqa: E402
from smoothmesh_amd.polymesh import cavity_mesh  # noqa: E402


def run(spec):
    kind, iters = spec.split(":")
    iters = int(iters)
    # a trailing "_off": constraints off -- the iteration then goes out as the two multi-role launches on tiles of the shared points
    # (k_geom_halo / k_smooth_halo, round 5); the engine says which form ran (`form` in the line)
    constraints = not kind.endswith("_off")
    kind = kind[:-4] if not constraints else kind
    if kind == "boxes":          # configs[1]/[2]'s family: 2 x 2 x 2 boxes of 32^3 cells, constraints on
        subs = [hex_subdomain((32, 32, 32), (2, 2, 2), r, jitter=0.2, seed=12345) for r in range(8)]
        what = "2x2x2 boxes of 32^3 hex cells"
    else:                        # configs[3]/[4]'s family: the castellated polyhedral mesh cut raggedly into five sub-domains
        gm = cavity_mesh(40, jitter=0.2, seed=12345)
        subs = decompose(gm, bfs_partition(gm, 5, seed=7, island=True), 5)
        what = f"polyhedral cavity mesh ({gm.nCells} cells), five breadth-first grown sub-domains, one of them disconnected"
    orcs = [oracle_ffi.Oracle(s.mesh) for s in subs]
    prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    for o in orcs:
        o.set_params(prm)
    mo = oracle_ffi.MultiOracle(orcs, *shared_point_table(subs))
    ms = LocalMultiSmoother(subs, device=0, overlap=False)
    ms.set_params(prm)
    chunk = max(1, iters // 10)
    done, bitwise, frozen_equal, t_or = 0, True, True, 0.0
    checkpoints = []
    t0 = time.perf_counter()
    while done < iters:
        k = min(chunk, iters - done)
        t1 = time.perf_counter()
        n_o, res_o, frz_o = mo.iterate(k, 0.0)
        t_or += time.perf_counter() - t1
        n_g, res_g, frz_g = ms.iterate(k, 0.0)
        done += k
        frozen_equal = frozen_equal and n_o == n_g and bool(np.array_equal(frz_o, frz_g))
        same = all(np.array_equal(p, o.points()) for p, o in zip(ms.get_points(), orcs))
        bitwise = bitwise and same
        checkpoints.append({"after": done, "bitwise_equal": bool(same), "nFrozenPoints": int(frz_g[-1]), "residual": float(res_g[-1])})
    hm = ms.states[0].eng.debug_halo_mode()
    return {"case": kind + ("" if constraints else "_off"), "what": what + (", constraints on" if constraints else ", constraints off"),
            "form": "multi-role launches" if hm["multi_role"] else "one kernel per step", "ranks": len(subs), "points": int(sum(s.mesh.nPoints for s in subs)), "iterations": iters,
            "compared_every": chunk, "bitwise_equal": bitwise, "nFrozen_series_equal": frozen_equal, "ok": bitwise and frozen_equal,
            "oracle_seconds": t_or, "wall_seconds": time.perf_counter() - t0, "checkpoints": checkpoints}


if __name__ == "__main__":
    bad = 0
    for spec in sys.argv[1:] or ["boxes:100", "irregular:200"]:
        r = run(spec)
        print(json.dumps(r), flush=True)
        bad += 0 if r["ok"] else 1
    sys.exit(1 if bad else 0)
