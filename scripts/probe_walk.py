import sys, time
sys.path.insert(0, '.')
import numpy as np
from smoothmesh_amd import SmoothEngine, default_params
from smoothmesh_amd.polymesh import cavity_mesh
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = cavity_mesh(N)
e = SmoothEngine(m)
p = default_params(e.mesh_stats()[0])
e.set_params(p)
e.debug_propose()
act = e.debug_field("faActive")
fr = e.debug_field("isFrozenPoint")
prop = e.debug_field("newPoints").reshape(-1, 3)
cur = e.debug_field("points").reshape(-1, 3)
moved = (prop != cur).any(axis=1)
print("points", m.nPoints, "active", int(act.sum()), "frozen", int(fr.sum()), "moved", int(moved.sum()), "active&moved", int((act.astype(bool) & moved).sum()))
e.enable_timing(True)
for i in range(3):
    e.debug_propose()
for c in e.counters():
    if c["launches"]:
        print(c["name"], c["ms"] / c["launches"] * 1e3, "us")
