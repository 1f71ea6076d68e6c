"""What one rank of an 8-rank (2x2x2) weak-scaling run does per iteration, on one GPU: the real sub-domain of rank 0
with its real halo tables (3 face neighbours, 3 edge neighbours, 1 corner neighbour), all device-side halo kernels and
split launches, and an RCCL self-exchange of the same volume standing in for the network.  Not a scaling number: it
isolates the compute-side cost of the multi-rank code path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29581")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from smoothmesh_amd import SmoothEngine, default_params
from smoothmesh_amd import halo
from smoothmesh_amd.meshgen import hex_subdomain
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
grid = (2, 2, 2)
subs = [hex_subdomain((n, n, n), grid, r, jitter=0.2, seed=12345) for r in range(8)]
cands = [s.processor_patch_point_lists() for s in subs]
sub = subs[0]
sub.nRanks = 1          # the process group has one member; the tables below are those of rank 0 among 8
real_gather = dist.all_gather_object
dist.all_gather_object = lambda out, obj: out.__setitem__(slice(None), [cands[0]])
class Fake(halo.DistributedSmoother):
    pass
t = halo.HaloTables(0, sub.pointProcAddressing, cands)
print(f"rank 0 of 8: {sub.mesh.nPoints} points, {len(t.sharedLocal)} shared, {t.nSend} send slots to {int((t.counts > 0).sum())} peers")
orig = halo.HaloTables
halo.HaloTables = lambda rank, ppa, c: t
modes = ("inorder", "overlap") if os.environ.get("SMOOTHMESH_EXCHANGE", "") != "push" else ("inorder",)
print("transport:", "peer stores (self-mapping: the rank's own receive slots and flag words stand in for its seven peers')"
      if len(modes) == 1 else "RCCL send / recv groups (self-exchange)")
for mode in modes:
    ds = halo.DistributedSmoother(sub, device=0, probe_slots=t.nSend, overlap=(mode == "overlap"))
    ds.set_params(default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False))
    ds.iterate(10, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ds.iterate(100, 0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"  {mode}: {1e4*dt:.1f} us/iter")
    if mode == "inorder":
        ds.engine.reset_counters(); ds.engine.enable_timing(True)
        ds.iterate(50, 0.0); torch.cuda.synchronize()
        ds.engine.enable_timing(False)
        for c in ds.engine.counters():
            if c["launches"]:
                print(f"      {c['name']:24s} launches/iter {c['launches']/50:.1f}  avg {1e3*c['ms']/c['launches']:.1f} us  per iter {1e3*c['ms']/50:.1f} us")
    del ds
if "--boundary" in sys.argv:
    # the same rank with boundary point smoothing: its three real sides onto the unit cube's surface (the other three sides are
    # processor patches); the L records (14 doubles per slot) travel through the same self-exchange
    from smoothmesh_amd import BoundaryParams
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    dist.all_gather_object = real_gather
    for mode in modes:
        ds = halo.DistributedSmoother(sub, device=0, probe_slots=t.nSend, overlap=(mode == "overlap"))
        prm = default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False)
        ds.set_params(prm)
        info = ds.set_boundary_smoothing(BoundaryParams(initEdges=box_feature_edges(n), targetSurfaces=box_surface(n // 2)), prm.minEdgeLength)
        ds.iterate(10, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ds.iterate(100, 0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"  boundary smoothing, {mode}: {1e4*dt:.1f} us/iter  ({info['nSmoothingSurfacePoints']} smoothing surface points)")
        del ds
e = SmoothEngine(sub.mesh, device=0)
e.set_params(default_params(e.mesh_stats()[0], edgeAngleConstraint=False, faceAngleConstraint=False))
e.iterate(10, 0.0)
t0 = time.perf_counter(); e.iterate(100, 0.0); dt = time.perf_counter() - t0
print(f"  same sub-domain as a serial mesh: {1e4*dt:.1f} us/iter")
dist.destroy_process_group()
