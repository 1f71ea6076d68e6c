"""What one rank of an 8-rank (2x2x2) weak-scaling run does per iteration, on one GPU: the real sub-domain of rank 0
with its real halo tables (3 face neighbours, 3 edge neighbours, 1 corner neighbour), all device-side halo kernels and
split launches, and an RCCL self-exchange of the same volume standing in for the network.  Not a scaling number: it
isolates the compute-side cost of the multi-rank code path.

usage: probe_rank_of_8.py [workload] [--boundary] [--iters K]
  workload = hexN[c] (default hex100): rank 0's N^3 block of the (2N)^3 hex block
           | cavityN[c]: box 0 of the castellated polyhedral mesh on a (2N)^3 base grid (cavity215c = BASELINE configs[4]'s rank:
             the 430^3-base, ~80 M-cell mesh; every box is generated once for its processor-patch point lists, one at a time)
  c = edgeAngle + faceAngle constraints on (minAngle 35 / maxAngle 160).
Prints us per iteration in order on the engine's stream, with the exchanges on the exchange stream, and for the same sub-domain
as a serial mesh (nothing packed, combined or exchanged), and weak_efficiency_bound = serial / best multi-rank arrangement."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29581")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import numpy as np  # noqa: E402

from smoothmesh_amd import SmoothEngine, default_params  # noqa: E402
from smoothmesh_amd import halo  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if args else "hex100"
if wl.isdigit():
    wl = "hex" + wl
K = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 100
constraints = wl.endswith("c")
base = wl[:-1] if constraints else wl
kind = "cavity" if base.startswith("cavity") else "hex"
n = int(base[len(kind):])
grid = (2, 2, 2)
t0 = time.perf_counter()
if kind == "hex":
    from smoothmesh_amd.meshgen import hex_subdomain
    subs = [hex_subdomain((n, n, n), grid, r, jitter=0.2, seed=12345) for r in range(8)]
    cands = [s.processor_patch_point_lists() for s in subs]
    sub = subs[0]
    del subs
else:
    from smoothmesh_amd.polymesh import cavity_subdomain
    cands = []
    sub = None
    for r in range(8):      # one box at a time: only the patch point lists of the seven peers are kept
        s = cavity_subdomain(2 * n, grid, r, jitter=0.2, seed=12345)
        cands.append(s.processor_patch_point_lists())
        if r == 0:
            sub = s
        del s
        gc.collect()
print(f"workload {wl}: rank 0 of 8 ({grid[0]}x{grid[1]}x{grid[2]}), constraints {'on (minAngle 35 / maxAngle 160)' if constraints else 'off'}; "
      f"sub-domains generated in {time.perf_counter() - t0:.1f} s", flush=True)
sub.nRanks = 1          # the process group has one member; the tables below are those of rank 0 among 8
real_gather = dist.all_gather_object
dist.all_gather_object = lambda out, obj: out.__setitem__(slice(None), [cands[0]])
t = halo.HaloTables(0, sub.pointProcAddressing, cands)
print(f"rank 0 of 8: {sub.mesh.nPoints} points, {sub.mesh.nCells} cells, {len(t.sharedLocal)} shared, {t.nSend} send slots to {int((t.counts > 0).sum())} peers")
halo.HaloTables = lambda rank, ppa, c: t
modes = ("inorder", "overlap") if os.environ.get("SMOOTHMESH_EXCHANGE", "") != "push" else ("inorder",)
print("transport:", "peer stores (self-mapping: the rank's own receive slots and flag words stand in for its seven peers')"
      if len(modes) == 1 else "RCCL send / recv groups (self-exchange)")
kw = dict(edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
us = {}
pts = {}
for mode in modes:
    ds = halo.DistributedSmoother(sub, device=0, probe_slots=t.nSend, overlap=(mode == "overlap"))
    prm = default_params(ds.global_min_edge(), **kw)
    ds.set_params(prm)
    hm = ds.engine.debug_halo_mode() if hasattr(ds.engine, "debug_halo_mode") else {}
    ds.iterate(10, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ds.iterate(K, 0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    us[mode] = 1e6 * dt / K
    pts[mode] = ds.engine.get_points()
    print(f"  {mode}: {us[mode]:.1f} us/iter   (iteration form: {'multi-role launches' if hm.get('multi_role') else 'one kernel per step'}"
          f"{', flagged' if hm.get('flagged') else ''})", flush=True)
    if mode == "inorder":
        ds.engine.reset_counters(); ds.engine.enable_timing(True)
        ds.iterate(50, 0.0); torch.cuda.synchronize()
        ds.engine.enable_timing(False)
        for c in ds.engine.counters():
            if c["launches"]:
                print(f"      {c['name']:24s} launches/iter {c['launches']/50:.1f}  avg {1e3*c['ms']/c['launches']:.1f} us  per iter {1e3*c['ms']/50:.1f} us")
    ds.close()
    del ds
if len(pts) == 2:
    print(f"  both arrangements leave the same coordinates after {10 + K} iterations: {bool(np.array_equal(pts['inorder'], pts['overlap']))}")
if "--boundary" in sys.argv and kind == "hex":
    # the same rank with boundary point smoothing: its three real sides onto the unit cube's surface (the other three sides are
    # processor patches); the L records (14 doubles per slot) travel through the same self-exchange
    from smoothmesh_amd import BoundaryParams
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    dist.all_gather_object = real_gather
    for mode in modes:
        ds = halo.DistributedSmoother(sub, device=0, probe_slots=t.nSend, overlap=(mode == "overlap"))
        prm = default_params(ds.global_min_edge(), **kw)
        ds.set_params(prm)
        info = ds.set_boundary_smoothing(BoundaryParams(initEdges=box_feature_edges(n), targetSurfaces=box_surface(n // 2)), prm.minEdgeLength)
        ds.iterate(10, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ds.iterate(100, 0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"  boundary smoothing, {mode}: {1e4*dt:.1f} us/iter  ({info['nSmoothingSurfacePoints']} smoothing surface points)")
        ds.close()
        del ds
e = SmoothEngine(sub.mesh, device=0)
e.set_params(default_params(e.mesh_stats()[0], **kw))
e.iterate(10, 0.0)
t0 = time.perf_counter(); e.iterate(K, 0.0); dt = time.perf_counter() - t0
serial = 1e6 * dt / K
best = min(us, key=us.get)
print(f"  same sub-domain as a serial mesh: {serial:.1f} us/iter")
print(f"  halo_overhead_us (best arrangement, {best}): {us[best] - serial:.1f}")
print(f"  weak_efficiency_bound = serial / multi-rank = {serial / us[best]:.4f}   (in order: {serial / us['inorder']:.4f})")
dist.destroy_process_group()
