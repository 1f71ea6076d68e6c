"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random block sizes, jitter up to near-inversion, random
smoothing parameters, constraints, layer patches, boundary point smoothing (box and sphere targets), serial and decomposed -- into
boxes, into IRREGULAR sub-domains (breadth-first grown or random cellRank maps, 2..8 ranks, disconnected pieces), and on
UNJITTERED / exactly graded blocks whose points sit on binary fractions, where the edge-length comparisons of the closest-point
syncs (SM.C:388-478) tie exactly; irregular hex cases also with boundary point smoothing, and with a baffle inside the block (shared
points or split twins, layers grown from it, ranks on its two sides) -- and ("affine") meshes in other units and far from the origin (scale 1e-6 .. 1e6, offset up to
1e7 mesh sizes: the f32 filters of the constraint evaluators work on f64 DIFFERENCES and must stay on the safe side, the
constraints are always on there).  Prints one line per case; exit code 1 on the first mismatch.  usage: python scripts/fuzz_parity.py [nCases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_ffi
from smoothmesh_amd import BoundaryParams, LayerParams, SmoothEngine, SmgpuError, default_params, patch_arrays
from smoothmesh_amd.surfgen import box_feature_edges, box_surface, sphere_surface
from smoothmesh_amd.decompose import bfs_partition, decompose, grid_partition, random_partition, shared_point_table
from smoothmesh_amd.halo import LocalMultiSmoother
from smoothmesh_amd.meshgen import add_baffle, baffle_in_plane, hex_block, hex_subdomain, split_baffles
from smoothmesh_amd.polymesh import cavity_mesh

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
KINDS = os.environ.get("FUZZ_KINDS", "hex,hex,cavity,multi,multi,irregular,tied").split(",")   # FUZZ_KINDS=irregular,tied: only those
PATCHES = ["xmin", "xmax", "ymin", "ymax", "zmin", "zmax"]


def params(mn, constraints_on=False):
    over = dict(edgeAngleConstraint=constraints_on or bool(rng.integers(2)), faceAngleConstraint=constraints_on or bool(rng.integers(2)),
                minAngle=float(rng.choice([15.0, 35.0, 50.0])), maxAngle=float(rng.choice([140.0, 160.0, 175.0])),
                relStepFrac=float(rng.choice([0.3, 0.5, 0.9])), totalMinFreeze=bool(rng.integers(2)))
    minEdge = float(rng.choice([0.3, 0.5, 0.9])) * mn
    return default_params(mn, minEdgeLength=minEdge, maxStepLength=float(rng.choice([0.1, 0.3, 0.6])) * minEdge, **over)


bad = 0
for case in range(n_cases):
    kind = rng.choice(KINDS)
    iters = int(rng.integers(2, 9))
    layers = rng.random() < 0.5
    lp = LayerParams(layerPatches=tuple(rng.choice(PATCHES, size=int(rng.integers(1, 4)), replace=False)),
                     layerMaxBlendingFraction=float(rng.choice([0.2, 0.3, 0.6])), layerExpansionRatio=float(rng.choice([1.0, 1.2, 1.5])),
                     minLayers=int(rng.integers(0, 3)), maxLayers=int(rng.integers(3, 6)))
    jitter = float(rng.choice([0.1, 0.3, 0.45]))
    seed = int(rng.integers(1 << 30))
    baffle = ""
    if kind in ("multi", "irregular", "tied"):
        if kind == "multi":
            grid = tuple(int(x) for x in rng.choice([1, 2, 3], size=3))
            if grid == (1, 1, 1):
                grid = (2, 1, 1)
            world = grid[0] * grid[1] * grid[2]
            nloc = tuple(int(x) for x in rng.integers(3, 9, size=3))
            subs = [hex_subdomain(nloc, grid, r, jitter=jitter, seed=seed) for r in range(world)]
            desc = f"multi grid {grid} local {nloc}"
        elif kind == "irregular":
            world = int(rng.integers(2, 9))
            if rng.random() < 0.5:
                dims = tuple(int(x) for x in rng.integers(5, 11, size=3))
                gm = hex_block(*dims, jitter=jitter, seed=seed)
                if rng.random() < 0.4:      # a wall inside the block (createBaffles), possibly split into twins (splitBaffles): ranks on its two sides
                    ax = int(rng.integers(3))
                    at = int(rng.integers(1, dims[ax])) / dims[ax]
                    lim = float(rng.choice([0.45, 0.7, 2.0]))
                    gm = add_baffle(gm, baffle_in_plane(hex_block(*dims), ax, at, lambda c: c[:, (ax + 1) % 3] < lim))
                    baffle = " baffle"
                    if rng.random() < 0.5:
                        gm = split_baffles(gm)
                        baffle = " split baffle"
                    if rng.random() < 0.5:
                        lp.layerPatches = ('"baffle.*"',) + tuple(lp.layerPatches[:1])
            else:
                gm = cavity_mesh(int(rng.integers(8, 13)), jitter=min(jitter, 0.3), seed=seed)
            how = rng.choice(["bfs", "island", "random"])
            cr = random_partition(gm, world, seed=seed) if how == "random" else bfs_partition(gm, world, seed=seed, island=(how == "island"))
            subs = decompose(gm, cr, world)
            desc = f"irregular {how} x{world} cells {gm.nCells}{baffle}"
        else:
            # points on binary fractions: spacing 2^-4 (x possibly 2^-5: an exactly graded block), a tenth of the interior points moved
            # by multiples of 2^-10 -- equal lengths stay bit-equal, so the closest-point syncs see exact ties at the processor cuts
            nx, ny, nz = (int(x) for x in rng.integers(4, 9, size=3))
            graded = rng.random() < 0.6
            if graded:
                nx *= 2
            gm = hex_block(nx, ny, nz, lengths=(nx / (32.0 if graded else 16.0), ny / 16.0, nz / 16.0), jitter=0.0)
            P = gm.points.reshape(-1, 3)
            inner = np.flatnonzero(gm.find_internal_points())
            mv = rng.choice(inner, size=max(1, len(inner) // 10), replace=False)
            P[mv] += rng.integers(-6, 7, size=(len(mv), 3)) / 1024.0
            grid = tuple(int(x) for x in rng.choice([1, 2], size=3))
            if grid == (1, 1, 1):
                grid = (2, 1, 1)
            world = grid[0] * grid[1] * grid[2]
            how = rng.choice(["boxes", "bfs"])
            cr = grid_partition(gm, grid) if how == "boxes" else bfs_partition(gm, world, seed=seed)
            subs = decompose(gm, cr, world)
            desc = f"tied {'graded ' if graded else ''}({nx},{ny},{nz}) {how} x{world}"
            jitter = 0.0
        ms = LocalMultiSmoother(subs, device=0)
        orcs = [oracle_ffi.Oracle(s.mesh) for s in subs]
        mn = min(o.mesh_stats()[0] for o in orcs)
        prm = params(mn)
        ms.set_params(prm)
        for o in orcs:
            o.set_params(prm)
        off, dom, loc = shared_point_table(subs)
        mo = oracle_ffi.MultiOracle(orcs, off, dom, loc)
        unit_hex = kind == "irregular" and "cavity" not in {p.name for p in subs[0].mesh.patches}   # the unit cube, cut raggedly
        boundary = (kind == "multi" or (unit_hex and not baffle)) and rng.random() < 0.5
        layers = layers and (kind != "irregular" or unit_hex)
        if boundary:      # multi: every sub-domain is a unit cube of the block [0, grid]
            hi = tuple(float(g) for g in grid) if kind == "multi" else (1.0, 1.0, 1.0)
            f = float(rng.choice([1.0, 1.0, 1.02]))
            c = 0.5 * np.array(hi)
            warp = (lambda x: c + (x - c) * f) if f != 1.0 else None
            m_e, m_s = int(rng.integers(1, 9)), int(rng.integers(1, 7))
            bp = BoundaryParams(initEdges=box_feature_edges(m_e, hi=hi), targetEdges=box_feature_edges(m_e, hi=hi, warp=warp) if warp else None,
                                targetSurfaces=box_surface(m_s, hi=hi, warp=warp), internalSmoothingBlendingFraction=float(rng.choice([0.0, 0.3, 1.0])))
            lopt = (lp.layerMaxBlendingFraction, prm.minEdgeLength, lp.layerExpansionRatio, lp.minLayers, lp.maxLayers)
            pa = [patch_arrays(s.mesh, lp.layerPatches if layers else ()) + (patch_arrays(s.mesh, bp.smoothingPatches)[3],) for s in subs]
            if layers:
                ms.set_layers(lp, prm.minEdgeLength)
            on_o = mo.setup_boundary(pa, lopt, bp.initEdges, bp.targetEdges, bp.targetSurfaces, bp.internalSmoothingBlendingFraction)
            assert on_o == bool(ms.set_boundary_smoothing(bp, prm.minEdgeLength)[0]["enabled"])
            desc += f" boundary->box*{f:g}"
        elif layers:
            on_o = mo.setup_layers([patch_arrays(s.mesh, lp.layerPatches) for s in subs], lp.layerMaxBlendingFraction, prm.minEdgeLength,
                                   lp.layerExpansionRatio, lp.minLayers, lp.maxLayers)
            assert on_o == ms.set_layers(lp, prm.minEdgeLength)
        err_o = err_g = None
        try:
            n_o, res_o, frz_o = mo.iterate(iters, 0.0)
        except RuntimeError as ex:
            err_o = str(ex)
        try:
            n_g, res_g, frz_g = ms.iterate(iters, 0.0)
        except SmgpuError as ex:
            err_g = str(ex)
        if err_o or err_g:
            ok = bool(err_o) and bool(err_g)
            print(f"case {case:3d} {'ok ' if ok else 'BAD'} {desc} jitter {jitter} iters {iters}: oracle error [{err_o}] engine error [{err_g}]", flush=True)
            bad += 0 if ok else 1
            continue
        a = np.concatenate(ms.get_points()); b = np.concatenate([o.points() for o in orcs])
    else:
        if kind == "hex":
            dims = tuple(int(x) for x in rng.integers(2, 14, size=3))
            mesh = hex_block(*dims, jitter=jitter, seed=seed)
            desc = f"hex {dims}"
        elif kind == "affine":
            if rng.random() < 0.5:
                dims = tuple(int(x) for x in rng.integers(3, 14, size=3))
                mesh = hex_block(*dims, jitter=jitter, seed=seed)
                desc = f"affine hex {dims}"
            else:
                n = int(rng.integers(8, 15))
                mesh = cavity_mesh(n, jitter=min(jitter, 0.3), seed=seed)
                desc = f"affine cavity {n}"
            scale = 10.0 ** float(rng.choice([-6, -3, 0, 2, 6]))
            shift = float(rng.choice([0.0, 1e2, 1e4, 1e6, 1e7])) * np.array([1.0, -0.7, 0.3])[rng.permutation(3)]
            P = mesh.points.reshape(-1, 3)
            P[:] = scale * (P + shift)
            desc += f" scale {scale:g} shift {np.max(np.abs(shift)):g}"
            layers = False
        else:
            n = int(rng.integers(8, 15))
            mesh = cavity_mesh(n, jitter=min(jitter, 0.3), seed=seed)
            lp.layerPatches = ("cavity",) if rng.random() < 0.5 else lp.layerPatches
            desc = f"cavity {n}"
        o = oracle_ffi.Oracle(mesh); e = SmoothEngine(mesh)
        prm = params(o.mesh_stats()[0], constraints_on=(kind == "affine"))
        o.set_params(prm); e.set_params(prm)
        boundary = kind != "affine" and rng.random() < 0.5
        st, sz, kd, sel = patch_arrays(mesh, lp.layerPatches if layers else ())
        lopt = (lp.layerMaxBlendingFraction, prm.minEdgeLength, lp.layerExpansionRatio, lp.minLayers, lp.maxLayers)
        if boundary:
            f = float(rng.choice([1.0, 1.0, 1.02]))
            warp = (lambda x: 0.5 + (x - 0.5) * f) if f != 1.0 else None
            m_e, m_s = int(rng.integers(1, 9)), int(rng.integers(1, 7))
            if kind == "hex":      # (kind "cavity" below; "affine" never gets here)
                pats = tuple(rng.choice(PATCHES, size=int(rng.integers(1, 7)), replace=False)) if rng.random() < 0.5 else ('".*"',)
                bp = BoundaryParams(initEdges=box_feature_edges(m_e), targetEdges=box_feature_edges(m_e, warp=warp) if warp else None,
                                    targetSurfaces=box_surface(m_s, warp=warp), smoothingPatches=pats,
                                    internalSmoothingBlendingFraction=float(rng.choice([0.0, 0.3, 1.0])))
            else:
                bp = BoundaryParams(initEdges=box_feature_edges(m_e), targetSurfaces=sphere_surface(levels=int(rng.integers(1, 5))),
                                    smoothingPatches=("cavity",), internalSmoothingBlendingFraction=float(rng.choice([0.0, 0.5])))
            desc += f" boundary->{'sphere' if kind != 'hex' else 'box*%g' % f} {bp.smoothingPatches}"
            if layers:
                e.set_layers(lp, prm.minEdgeLength)
            on_o = o.setup_boundary(st, sz, kd, sel, patch_arrays(mesh, bp.smoothingPatches)[3], lopt, bp.initEdges, bp.targetEdges,
                                    bp.targetSurfaces, None, None, bp.internalSmoothingBlendingFraction)
            assert on_o == bool(e.set_boundary_smoothing(bp, prm.minEdgeLength)["enabled"])
        elif layers:
            on_o = o.setup_layers(st, sz, kd, sel, *lopt)
            assert on_o == e.set_layers(lp, prm.minEdgeLength)
        err_o = err_g = None
        try:
            n_o, res_o, frz_o = o.iterate(iters, 0.0)
        except RuntimeError as ex:
            err_o = str(ex)
        try:
            n_g, res_g, frz_g = e.iterate(iters, 0.0)
        except SmgpuError as ex:
            err_g = str(ex)
        if err_o or err_g:      # the reference's FatalErrors (e.g. no surface intersection): both sides must report one
            ok = bool(err_o) and bool(err_g)
            print(f"case {case:3d} {'ok ' if ok else 'BAD'} {desc} jitter {jitter} iters {iters}: oracle error [{err_o}] engine error [{err_g}]", flush=True)
            bad += 0 if ok else 1
            continue
        a, b = e.get_points(), o.points()
    fin = ~np.isnan(b)
    ok = (n_o == n_g and np.array_equal(frz_o, frz_g) and np.array_equal(np.isnan(a), np.isnan(b)) and
          (not fin.any() or np.max(np.abs(a[fin] - b[fin])) <= 1e-13 * (max(1.0, np.max(np.abs(b[fin]))) if kind != "affine" else np.max(np.abs(b[fin])))))
    print(f"case {case:3d} {'ok ' if ok else 'BAD'} {desc} jitter {jitter} iters {iters} layers {lp.layerPatches if layers else '-'} "
          f"ea {prm.edgeAngleConstraint} fa {prm.faceAngleConstraint} frozen {frz_o[-1]} maxdiff {np.max(np.abs(a[fin] - b[fin])) if fin.any() else 0:.2e}", flush=True)
    bad += 0 if ok else 1
sys.exit(1 if bad else 0)
