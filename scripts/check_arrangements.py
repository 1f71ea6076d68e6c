"""GPU check that every arrangement of the multi-rank iteration gives the same bits: one rank of eight (2 x 2 x 2 cut of a hex
block: face, edge and corner sharers) with its real halo tables and a self-exchange standing in for the seven peers (RCCL send /
recv groups with the one rank a 1-GPU box offers; SMOOTHMESH_EXCHANGE=push: the peer-store transport onto the rank's own receive
slots) -- the one-kernel-per-step form (SMGPU_HALO_MERGED=0) in order and with an exchange stream, the multi-role launches
(k_geom_halo / k_smooth_halo) in order, and their flagged arrangement (exchanges ordered by flag words next to the launches).
With the self-exchange a shared point is combined with its own record, so the result is no mesh anybody wants -- but it is a fixed
function of the inputs that every arrangement must reproduce bit for bit.  Prints 'arrangements: ok' or exits non-zero."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29589")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from smoothmesh_amd import default_params
from smoothmesh_amd import halo
from smoothmesh_amd.meshgen import hex_subdomain
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 14
grid = (2, 2, 2)
subs = [hex_subdomain((n, n, n), grid, r, jitter=0.25, seed=77) for r in range(8)]
cands = [s.processor_patch_point_lists() for s in subs]
sub = subs[0]
sub.nRanks = 1
dist.all_gather_object = lambda out, obj: out.__setitem__(slice(None), [cands[0]])
t = halo.HaloTables(0, sub.pointProcAddressing, cands)
halo.HaloTables = lambda rank, ppa, c: t
push = os.environ.get("SMOOTHMESH_EXCHANGE", "") == "push"
cases = [("one kernel per step, in order", {"SMGPU_HALO_MERGED": "0"}, False),
         ("multi-role launches, in order", {}, False)]
if not push:
    cases += [("one kernel per step, exchange stream", {"SMGPU_HALO_MERGED": "0"}, True),
              ("exchange stream without the flag words (SMGPU_HALO_FLAGGED=0: back to one kernel per step)", {"SMGPU_HALO_FLAGGED": "0"}, True),
              ("multi-role launches, flagged", {}, True)]
ref = None
bad = 0
for name, env, overlap in cases:
    for k in ("SMGPU_HALO_MERGED", "SMGPU_HALO_FLAGGED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ds = halo.DistributedSmoother(sub, device=0, probe_slots=t.nSend, overlap=overlap)
    ds.set_params(default_params(ds.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False))
    done, res, frz = ds.iterate(iters, 0.0)
    pts = ds.get_points()
    hm = ds.engine.debug_halo_mode()
    want = {"multi_role": "multi-role" in name, "flagged": "flagged" in name}
    if (hm["multi_role"], hm["flagged"]) != (want["multi_role"], want["flagged"]):
        print(f"{name}: the engine took another path: {hm}")
        bad += 1
    ds.close()
    got = (done, res.copy(), frz.copy(), pts.copy())
    if ref is None:
        ref = got
        print(f"{name}: reference ({t.nSend} send slots, {len(t.sharedLocal)} shared points, nFrozenPoints {frz[:4].tolist()} ..., residual {res[-1]:.6g})")
        continue
    same = got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    print(f"{name}: {'same bits' if same else 'DIFFERENT: max |dx| = %.3e' % float(np.max(np.abs(got[3] - ref[3])))}")
    bad += 0 if same else 1
dist.destroy_process_group()
if bad:
    sys.exit(1)
print("arrangements: ok")
