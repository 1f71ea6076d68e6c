"""Which threads of a bench run burn CPU (diagnosis of host-side stalls): run a workload, then list per-thread CPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_mesh, parse_workload
from smoothmesh_amd import SmoothEngine, default_params
wl = sys.argv[1] if len(sys.argv) > 1 else "cavity100c"
kind, n, con = parse_workload(wl)
mesh = make_mesh(kind, n)
eng = SmoothEngine(mesh, device=0)
eng.set_params(default_params(eng.mesh_stats()[0], edgeAngleConstraint=con, faceAngleConstraint=con))
eng.iterate(3, 0.0)
def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
        out[t] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
    return out
a = snap(); t0 = time.perf_counter()
eng.iterate(20, 0.0)
dt = time.perf_counter() - t0; b = snap()
print(f"wall {dt:.3f} s, threads {len(b)}")
for t, v in sorted(b.items(), key=lambda kv: -(kv[1] - a.get(kv[0], 0)))[:8]:
    name = open(f"/proc/self/task/{t}/comm").read().strip()
    print(f"  tid {t} {name:20s} cpu {v - a.get(t, 0):.3f} s")
print("cgroup cpu.stat:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " ") if os.path.exists("/sys/fs/cgroup/cpu.stat") else "n/a")
