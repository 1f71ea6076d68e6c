"""DistributedSmoother on a decomposed POLYHEDRAL mesh (BASELINE configs[4]'s family) against the oracle's MultiDomain, on N
ranks (torch.distributed.run); every rank generates its own sub-domain (polymesh.cavity_subdomain).
On a 1-GPU box: SMOOTHMESH_SHARE_GPU=1 SMOOTHMESH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... (the
engines are the real ones; only the transport is gloo).  CHECK_IRREGULAR=<kind>:<seed> takes the sub-domains from an IRREGULAR
cellRank instead (tests/test_irregular_partitions.build_case: ragged interfaces, a disconnected sub-domain, a rank without shared
points; any world size).  Exit code 1 on a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, torch.distributed as dist
from oracle import oracle_ffi
from smoothmesh_amd import default_params
from smoothmesh_amd.decompose import shared_point_table
from smoothmesh_amd.halo import DistributedSmoother
from smoothmesh_amd.polymesh import cavity_subdomain

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0"))
if os.environ.get("SMOOTHMESH_SHARE_GPU"):
    local %= torch.cuda.device_count()
torch.cuda.set_device(local)
backend = os.environ.get("SMOOTHMESH_BACKEND", "nccl")
dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
irregular = os.environ.get("CHECK_IRREGULAR", "")
grid = None if irregular else {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}[world]
N = int(os.environ.get("CHECK_POLY_N", "14"))


def make_subs():
    if irregular:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_irregular_partitions import build_case
        from smoothmesh_amd.decompose import decompose
        kind, seed = irregular.split(":")
        mesh, cr = build_case(kind, world, int(seed))
        return decompose(mesh, cr, world)
    return [cavity_subdomain(N, grid, r, jitter=0.2, seed=4) for r in range(world)]


bad = 0
for constraints in (False, True):
    for overlap in (False, True):
        subs = make_subs()   # all of them only for the expected values
        orcs = [oracle_ffi.Oracle(s.mesh) for s in subs]
        prm = default_params(min(o.mesh_stats()[0] for o in orcs), edgeAngleConstraint=constraints, faceAngleConstraint=constraints)
        for o in orcs:
            o.set_params(prm)
        mo = oracle_ffi.MultiOracle(orcs, *shared_point_table(subs))
        ds = DistributedSmoother(subs[rank], device=local, overlap=overlap)
        assert ds.global_min_edge() == min(o.mesh_stats()[0] for o in orcs)
        ds.set_params(prm)
        n_o, res_o, frz_o = mo.iterate(7, 0.0)
        n_g, res_g, frz_g = ds.iterate(7, 0.0)
        diff = float(np.max(np.abs(ds.engine.get_points() - orcs[rank].points())))
        ok = n_o == n_g and np.array_equal(np.asarray(frz_o), np.asarray(frz_g)) and diff <= 1e-13
        transport = "peer stores" if ds.pushbuf is not None else ("send/recv groups" if ds.direct is not None else backend)
        hm = ds.engine.debug_halo_mode()
        transport += ", " + ("multi-role launches" + (", flagged" if hm["flagged"] else "") + (", fix inside" if hm["fix_inside"] else "") if hm["multi_role"] else "one kernel per step")
        print(f"rank {rank} constraints {constraints} overlap {overlap} [{transport}]: {'ok' if ok else 'BAD'} max diff {diff:.2e} frozen {list(frz_g)[-1]}", flush=True)
        bad += 0 if ok else 1
        ds.close()
        del ds
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
