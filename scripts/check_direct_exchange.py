"""GPU check of the direct RCCL exchange of the Python driver (smoothmesh_amd/rccl_direct.py) with the one rank a 1-GPU box
offers: the communicator comes up beside torch's, the start-up self-check against all_to_all_single passes, and a loop that
issues the send / recv groups on the engine's stream (a self-exchange of dummy records: world = 1 has no shared points)
gives the serial loop's coordinates and per-iteration records.  Prints 'direct exchange: ok' or exits non-zero."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29587")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from smoothmesh_amd import SmoothEngine, default_params
from smoothmesh_amd.halo import DistributedSmoother
from smoothmesh_amd.meshgen import hex_block, hex_subdomain
bad = 0
for overlap in (False, True):
    sub = hex_subdomain((14, 12, 10), (1, 1, 1), 0, jitter=0.3, seed=5)
    ds = DistributedSmoother(sub, device=0, overlap=overlap, probe_slots=257)
    if ds.direct is None:
        print("the direct exchange did not come up (self-check failed or library missing)"); bad += 1; continue
    prm = default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False)
    ds.set_params(prm)
    n_d, res_d, frz_d = ds.iterate(9, 0.0)
    eng = SmoothEngine(hex_block(14, 12, 10, lengths=(1.0, 1.0, 1.0), jitter=0.3, seed=5), device=0)
    eng.set_params(prm)
    n_s, res_s, frz_s = eng.iterate(9, 0.0)
    same = n_d == n_s and np.array_equal(frz_d, frz_s) and np.array_equal(res_d, res_s) and np.array_equal(ds.get_points(), eng.get_points())
    print(f"overlap={overlap}: {'same' if same else 'DIFFERENT'}")
    bad += 0 if same else 1
    del ds
dist.destroy_process_group()
if bad:
    sys.exit(1)
print("direct exchange: ok")
