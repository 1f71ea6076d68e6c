import sys
sys.path.insert(0, '.')
import numpy as np, torch
from smoothmesh_amd import default_params
from smoothmesh_amd.halo import LocalMultiSmoother
from smoothmesh_amd.meshgen import hex_subdomain
grid = (2, 1, 1)
subs = [hex_subdomain((5, 4, 4), grid, r, jitter=0.3, seed=9) for r in range(2)]
ms = LocalMultiSmoother(subs, device=0)
prm = default_params(ms.global_min_edge(), edgeAngleConstraint=False, faceAngleConstraint=False)
ms.set_params(prm)
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
for st in ms.states: st.eng.iter_begin()
sync("begin")
for st in ms.states: st.eng.iter_interior()
sync("interior")
ms._exchange("A"); sync("xA")
for st in ms.states: st.eng.iter_mid()
sync("mid")
ms._exchange("F"); sync("xF")
for st in ms.states: st.eng.iter_end()
sync("end")
