"""Kernel time of the segment query alone (k_bnd_find_line under rocprofv3): the rays k_bnd_fix traces on the hex100B
workload -- from 0.09 h inside the block surface, 0.3 h long, outwards and inwards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smoothmesh_amd import BoundaryParams, SmoothEngine, default_params
from smoothmesh_amd.meshgen import hex_block
from smoothmesh_amd.surfgen import box_feature_edges, box_surface
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = hex_block(n, jitter=0.2, seed=12345)
e = SmoothEngine(m)
prm = default_params(e.mesh_stats()[0], edgeAngleConstraint=False, faceAngleConstraint=False)
e.set_params(prm)
e.set_boundary_smoothing(BoundaryParams(initEdges=box_feature_edges(n), targetSurfaces=box_surface(n // 2)), prm.minEdgeLength)
p = np.array(m.points)
h = 1.0 / n
starts, ends = [], []
for a in range(3):
    for side, sgn in ((0.0, 1.0), (1.0, -1.0)):
        q = p[p[:, a] == side].copy()
        q[:, a] += sgn * 0.09 * h
        for d in (1.0, -1.0):
            r = q.copy(); r[:, a] += d * 0.3 * h
            starts.append(q); ends.append(r)
starts, ends = np.concatenate(starts), np.concatenate(ends)
for _ in range(5):
    hit, pts = e.debug_find_line(starts, ends)
print(len(starts), "segments,", int(hit.sum()), "hits")
