"""Shared builders for the boundary point smoothing tests (oracle and GPU sides get the same inputs)."""
import numpy as np


def scale_about_centre(f):
    return lambda x: 0.5 + (x - 0.5) * f


def tangential_jitter(mesh, amp, seed):
    """moves the boundary points of a unit-cube block inside their face / along their edge (corners stay)"""
    rng = np.random.default_rng(seed)
    p = np.array(mesh.points, np.float64).copy()
    on_lo, on_hi = p == 0.0, p == 1.0
    fixed = on_lo | on_hi                                   # per coordinate: pinned to a box plane
    bnd = fixed.any(axis=1)
    d = rng.uniform(-amp, amp, p.shape)
    d[fixed] = 0.0
    p[bnd] += d[bnd]
    mesh.points[:] = p
    return mesh


def boundary_inputs(n_edge_segments, n_surf, warp=None):
    from smoothmesh_amd.surfgen import box_feature_edges, box_surface
    init = box_feature_edges(n_edge_segments)
    target = box_feature_edges(n_edge_segments, warp=warp) if warp is not None else None
    surf = box_surface(n_surf, warp=warp)
    return init, target, surf


def make_pair(mesh, oracle_lib, init, target, surf, constraints=False, layerPatches=(), engine=True, blend=0.0,
              smoothingPatches=('".*"',), cornerIO=None, featureIO=None, **prm_over):
    """-> (oracle, engine or None, params, doBoundarySmoothing)"""
    from smoothmesh_amd import BoundaryParams, LayerParams, SmoothEngine, default_params, patch_arrays
    o = oracle_lib.Oracle(mesh)
    prm = default_params(o.mesh_stats()[0], edgeAngleConstraint=constraints, faceAngleConstraint=constraints, **prm_over)
    o.set_params(prm)
    st, sz, kd, sel_l = patch_arrays(mesh, layerPatches)
    sel_s = patch_arrays(mesh, smoothingPatches)[3]
    L = LayerParams(layerPatches=tuple(layerPatches))
    on_o = o.setup_boundary(st, sz, kd, sel_l, sel_s, (L.layerMaxBlendingFraction, prm.minEdgeLength, L.layerExpansionRatio,
                                                       L.minLayers, L.maxLayers), init, target, surf, cornerIO, featureIO, blend)
    e = None
    if engine:
        e = SmoothEngine(mesh)
        e.set_params(prm)
        if layerPatches:
            e.set_layers(L, prm.minEdgeLength)
        info = e.set_boundary_smoothing(BoundaryParams(initEdges=init, targetSurfaces=surf, targetEdges=target,
                                                       smoothingPatches=tuple(smoothingPatches),
                                                       internalSmoothingBlendingFraction=blend, isCornerPointIO=cornerIO,
                                                       isFeatureEdgePointIO=featureIO), prm.minEdgeLength)
        assert bool(info["enabled"]) == on_o
        e.boundary_info = info
    return o, e, prm, on_o
