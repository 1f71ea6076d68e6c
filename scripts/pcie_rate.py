"""PCIe-inclusive rates for DESIGN.md: coordinates in (set_points) + K iterations + coordinates out (get_points), and the
one-off upload of the mesh tables at smgpu_create."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_mesh, parse_workload
from smoothmesh_amd import SmoothEngine, default_params
wl = sys.argv[1] if len(sys.argv) > 1 else "hex100"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kind, n, con = parse_workload(wl)
mesh = make_mesh(kind, n)
t0 = time.perf_counter(); eng = SmoothEngine(mesh, device=0); t_create = time.perf_counter() - t0
eng.set_params(default_params(eng.mesh_stats()[0], edgeAngleConstraint=con, faceAngleConstraint=con))
eng.iterate(5, 0.0)
pts = np.ascontiguousarray(mesh.points)
for rep in range(3):
    t0 = time.perf_counter(); eng.set_points(pts); t1 = time.perf_counter()
    eng.iterate(K, 0.0); t2 = time.perf_counter()
    out = eng.get_points(); t3 = time.perf_counter()
print(f"{wl}: create (addressing + tiles + upload of {eng.sizes()['deviceBytes']/1e9:.2f} GB) {t_create:.2f} s; "
      f"set_points {1e3*(t1-t0):.2f} ms, {K} iterations {1e3*(t2-t1):.2f} ms, get_points {1e3*(t3-t2):.2f} ms -> "
      f"{mesh.nPoints*K/(t3-t0)/1e9:.3f} Gpts/s with the coordinates crossing PCIe both ways, {mesh.nPoints*K/(t2-t1)/1e9:.3f} resident")
