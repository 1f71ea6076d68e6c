"""N=1 probe of the multi-rank iteration's fixed costs: the exchange machinery (RCCL self-exchange of dummy
records through the same streams/events as a real run) for several record counts and overlap modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29579")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from smoothmesh_amd import default_params
from smoothmesh_amd.halo import DistributedSmoother
from smoothmesh_amd.meshgen import hex_subdomain
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sub = hex_subdomain((n, n, n), (1, 1, 1), 0, jitter=0.2, seed=12345)
for slots in (0, 1000, 30000):
    for mode in ("overlap", "inorder"):
        ds = DistributedSmoother(sub, device=0, probe_slots=slots, overlap=(mode == "overlap"))
        ds.set_params(default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False))
        ds.iterate(10, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.iterate(100, 0.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if mode == "overlap" and slots:
            print("   autotune:", ds.autotune(10), flush=True)
        print(f"slots={slots} mode={mode}: host {1e4*(t1-t0):.1f} us/iter, total {1e4*(t2-t0):.1f} us/iter", flush=True)
        del ds
dist.destroy_process_group()
