"""per-iteration wall time of the constraints-on loop on the polyhedral cavity mesh (face-angle walk statistics with SMGPU_VERBOSE=2)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from smoothmesh_amd import SmoothEngine, default_params
from smoothmesh_amd.polymesh import cavity_mesh
N = int(sys.argv[1]); it = int(sys.argv[2])
m = cavity_mesh(N)
e = SmoothEngine(m)
e.set_params(default_params(e.mesh_stats()[0]))
for i in range(it):
    t = time.time(); n, res, frz = e.iterate(1, 0.0); print(f"iteration {i}: {(time.time() - t) * 1e3:.2f} ms, nFrozenPoints {frz[0]}", flush=True)
