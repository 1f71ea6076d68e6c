"""How many of the walk predicates' stars have, bit for bit, the inputs they had in the point's previous walk (SMGPU_WALK_MEMO_STATS=1,
k_walk_pred_pack: vertex slots with roles, entries' proposals and states, the point's two positions and angle bounds, the ring places'
cell centres)?  Needs a measuring build of the library (the hash block is not in the product kernel):
    make -C smoothmesh_amd/csrc HIPFLAGS+=-DSMGPU_WALK_MEMO=1 SMGPU_LIB=/tmp/libsmgpu_memo.so  ->  SMOOTHMESH_SMGPU_LIB=/tmp/libsmgpu_memo.so
Decides whether an exact memo of the predicates could pay.  usage: walk_memo_stats.py [workload] [chunks of 10 iterations]"""
import os, sys
os.environ["SMGPU_WALK_MEMO_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_mesh, parse_workload
from smoothmesh_amd import SmoothEngine, default_params
wl = sys.argv[1] if len(sys.argv) > 1 else "cavity100c"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind, n, con = parse_workload(wl)
eng = SmoothEngine(make_mesh(kind, n), device=0)
eng.set_params(default_params(eng.mesh_stats()[0], edgeAngleConstraint=True, faceAngleConstraint=True))
done = 0
for c in range(chunks):
    n_, res, frz = eng.iterate(10, 0.0)
    done += 10
    print(f"{wl}: after {done} iterations nFrozenPoints {int(frz[-1])} residual {float(res[-1]):.4g}", file=sys.stderr, flush=True)
# sanity of the hash itself: the SAME iteration twice from the same coordinates must match for every star
p0 = eng.get_points()
eng.iterate(1, 0.0)
eng.set_points(p0)
print(f"{wl}: control -- the next line repeats the last iteration from the same coordinates: expect 100 %", file=sys.stderr, flush=True)
eng.iterate(1, 0.0)
