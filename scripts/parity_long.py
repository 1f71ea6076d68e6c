#!/usr/bin/env python3
"""Long-run parity: the engine against the CPU oracle over BASELINE.json's own iteration counts (or as many as the oracle's cost
allows), compared every `chunk` iterations: identical nFrozenPoints series, coordinates within 1e-10 relative L-inf (north star;
measured: bit-equal).  One JSON line per workload on stdout; run on the GPU box:
    python scripts/parity_long.py hex100:100 hex100c:100 cavity100c:200 cavity215c:20
The oracle is test infrastructure (oracle/); nothing here is timed or shipped."""
import json
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload names and meshes as bench.py builds them)
from oracle import oracle_ffi  # noqa: E402
from smoothmesh_amd import SmoothEngine, default_params  # noqa: E402


def run(spec):
    wl, iters = spec.split(":")
    iters = int(iters)
    kind, n, constraints = bench.parse_workload(wl)
    mesh = bench.make_mesh(kind, n)
    t0 = time.perf_counter()
    o = oracle_ffi.Oracle(mesh)
    e = SmoothEngine(mesh)
    p = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
    o.set_params(p); e.set_params(p)
    chunk = max(1, iters // 10)
    done, worst, frozen_equal, bitwise = 0, 0.0, True, True
    t_or = 0.0
    series = []
    while done < iters:
        k = min(chunk, iters - done)
        t1 = time.perf_counter()
        n_o, res_o, frz_o = o.iterate(k, 0.0)
        t_or += time.perf_counter() - t1
        n_g, res_g, frz_g = e.iterate(k, 0.0)
        done += k
        a, b = e.get_points(), o.points()
        err = float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        worst = max(worst, err)
        frozen_equal = frozen_equal and n_o == n_g == k and bool(np.array_equal(frz_o, frz_g))
        bitwise = bitwise and bool(np.array_equal(a, b))
        series.append({"after": done, "rel_linf": err, "nFrozenPoints": int(frz_g[-1]), "residual": float(res_g[-1])})
    mode, switches = e.debug_walk_mode()
    out = {"workload": wl, "points": int(mesh.nPoints), "cells": int(mesh.nCells), "iterations": iters, "compared_every": chunk,
           "rel_linf_max": worst, "bitwise_equal": bitwise, "nFrozen_series_equal": frozen_equal, "tolerance": 1e-10,
           "ok": bool(worst <= 1e-10 and frozen_equal), "walk_replay_form_last": mode, "walk_replay_form_changes": switches,
           "oracle_seconds": t_or, "wall_seconds": time.perf_counter() - t0, "checkpoints": series}
    e.close(); o.close()
    print(json.dumps(out), flush=True)
    return out["ok"]


if __name__ == "__main__":
    ok = all([run(s) for s in sys.argv[1:]])
    sys.exit(0 if ok else 1)
