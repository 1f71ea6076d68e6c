"""DistributedSmoother with boundary point smoothing against the oracle's MultiDomain, on N ranks (torch.distributed.run).
On a 1-GPU box: SMOOTHMESH_SHARE_GPU=1 SMOOTHMESH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... (the
engines are the real ones; only the transport is gloo).  Exit code 1 on a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist
from oracle import oracle_ffi
from smoothmesh_amd import BoundaryParams, LayerParams
from smoothmesh_amd.halo import DistributedSmoother
from smoothmesh_amd.surfgen import box_feature_edges, box_surface
from test_oracle_boundary import _multi_boundary_case

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0"))
if os.environ.get("SMOOTHMESH_SHARE_GPU"):
    local %= torch.cuda.device_count()
torch.cuda.set_device(local)
backend = os.environ.get("SMOOTHMESH_BACKEND", "nccl")
dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
grid = {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}[world]
bad = 0
for constraints, layers in ((False, False), (True, True)):
    lpatches = ("xmin", "zmax") if layers else ()
    mo, orcs, subs, _, hi = _multi_boundary_case(oracle_ffi, grid, (6, 5, 4), 0.25, constraints, blend=0.4, layerPatches=lpatches)
    ds = DistributedSmoother(subs[rank], device=local)
    ds.set_params(mo.params)
    if layers:
        assert ds.set_layers(LayerParams(layerPatches=lpatches), mo.params.minEdgeLength)
    info = ds.set_boundary_smoothing(BoundaryParams(initEdges=box_feature_edges(8, hi=hi), targetSurfaces=box_surface(4, hi=hi),
                                                    internalSmoothingBlendingFraction=0.4), mo.params.minEdgeLength)
    assert info["enabled"]
    n_o, res_o, frz_o = mo.iterate(7, 0.0)
    n_g, res_g, frz_g = ds.iterate(7, 0.0)
    diff = float(np.max(np.abs(ds.engine.get_points() - orcs[rank].points())))
    ok = n_o == n_g and np.array_equal(np.asarray(frz_o), np.asarray(frz_g)) and diff <= 1e-13
    print(f"rank {rank} constraints {constraints} layers {layers}: {'ok' if ok else 'BAD'} max diff {diff:.2e} frozen {list(frz_g)[-1]}", flush=True)
    bad += 0 if ok else 1
    ds.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
