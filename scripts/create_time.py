"""Host-side set-up cost of one engine (addressing + tile tables + upload) for a workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_mesh, parse_workload
from smoothmesh_amd import SmoothEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "hex100"
kind, n, con = parse_workload(wl)
t0 = time.perf_counter(); mesh = make_mesh(kind, n); t1 = time.perf_counter()
if "--cold" not in sys.argv:      # the HIP runtime's own first-use work (queues, copy / fill kernels: ~0.2 s per process) is not the engine's set-up
    torch.zeros(1 << 20, device="cuda").cpu(); torch.cuda.synchronize()
    t1 = time.perf_counter()
eng = SmoothEngine(mesh, device=0); t2 = time.perf_counter()
gb = eng.sizes()["deviceBytes"] / 1e9
print(f"{wl}: mesh generation {t1-t0:.2f} s, find_internal_points+smgpu_create {t2-t1:.2f} s, points {mesh.nPoints}, device bytes {gb:.2f} GB")
