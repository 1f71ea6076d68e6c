"""Synthetic polyMesh generators standing in for OpenFOAM's blockMesh / decomposePar (unavailable here).

hex_block() reproduces blockMesh's single-block numbering: point id
i + j(nx+1) + k(nx+1)(ny+1), cell id i + j nx + k nx ny, internal faces in upper-triangular order
(per cell: +x, +y, +z neighbour), then the six boundary patches.  Interior points can be jittered so
that smoothing has work to do (a uniform block is a fixed point, src/smoothMesh.C:2401).  The jitter
is a counter-based hash of (seed, global point id, axis), so a sub-domain generated directly by
hex_subdomain() carries exactly the coordinates the global mesh would give it.
"""
import numpy as np

from .mesh import PolyMesh, Patch


def _hash_uniform(seed, gid, axis):
    """splitmix64 of (seed, 3*gid+axis) -> uniform [0,1) doubles; vectorised over gid."""
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + (gid.astype(np.uint64) * np.uint64(3) + np.uint64(axis))
             * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _lattice_points(gi, gj, gk, GN, lengths, jitter, seed):
    """coordinates of lattice points (global indices gi,gj,gk) of a GN=(GNX,GNY,GNZ)-cell block"""
    GNX, GNY, GNZ = GN
    hx, hy, hz = lengths[0] / GNX, lengths[1] / GNY, lengths[2] / GNZ
    pts = np.stack([gi * hx, gj * hy, gk * hz], axis=1).astype(np.float64)
    if jitter > 0.0:
        gid = gi.astype(np.int64) + gj.astype(np.int64) * (GNX + 1) + gk.astype(np.int64) * (GNX + 1) * (GNY + 1)
        interior = (gi > 0) & (gi < GNX) & (gj > 0) & (gj < GNY) & (gk > 0) & (gk < GNZ)
        h = np.array([hx, hy, hz])
        for a in range(3):
            u = _hash_uniform(seed, gid, a)
            pts[:, a] += np.where(interior, (2.0 * u - 1.0) * jitter * h[a], 0.0)
    return pts


def hex_block(nx, ny=None, nz=None, lengths=(1.0, 1.0, 1.0), jitter=0.0, seed=12345) -> PolyMesh:
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    npx, npy = nx + 1, ny + 1

    def pid(i, j, k):
        return (i + j * npx + k * npx * npy).astype(np.int64)

    K, J, I = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    pts = _lattice_points(I.ravel(), J.ravel(), K.ravel(), (nx, ny, nz), lengths, jitter, seed)

    ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    cid = ci + cj * nx + ck * nx * ny

    def quad_x(i, j, k):  # normal +x
        return np.stack([pid(i, j, k), pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i, j, k + 1)], axis=1)

    def quad_y(i, j, k):  # normal +y
        return np.stack([pid(i, j, k), pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j, k)], axis=1)

    def quad_z(i, j, k):  # normal +z
        return np.stack([pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i, j + 1, k)], axis=1)

    # internal faces: cell-major, (+x, +y, +z) within a cell
    fx = quad_x(ci + 1, cj, ck); vx = ci < nx - 1
    fy = quad_y(ci, cj + 1, ck); vy = cj < ny - 1
    fz = quad_z(ci, cj, ck + 1); vz = ck < nz - 1
    faces3 = np.stack([fx, fy, fz], axis=1)                  # (C, 3, 4)
    valid3 = np.stack([vx, vy, vz], axis=1)                  # (C, 3)
    nei3 = np.stack([cid + 1, cid + nx, cid + nx * ny], axis=1)
    own3 = np.repeat(cid[:, None], 3, axis=1)
    int_faces = faces3[valid3]
    int_own = own3[valid3]
    int_nei = nei3[valid3]

    def sel(mask):
        return ci[mask], cj[mask], ck[mask], cid[mask]

    bfaces, bown, patches = [], [], []
    start = len(int_faces)

    def add_patch(name, quads, owners):
        nonlocal start
        bfaces.append(quads); bown.append(owners)
        patches.append(Patch(name=name, type="patch", nFaces=len(quads), startFace=start))
        start += len(quads)

    i, j, k, c = sel(ci == 0)
    add_patch("xmin", quad_x(i, j, k)[:, ::-1], c)
    i, j, k, c = sel(ci == nx - 1)
    add_patch("xmax", quad_x(i + 1, j, k), c)
    i, j, k, c = sel(cj == 0)
    add_patch("ymin", quad_y(i, j, k)[:, ::-1], c)
    i, j, k, c = sel(cj == ny - 1)
    add_patch("ymax", quad_y(i, j + 1, k), c)
    i, j, k, c = sel(ck == 0)
    add_patch("zmin", quad_z(i, j, k)[:, ::-1], c)
    i, j, k, c = sel(ck == nz - 1)
    add_patch("zmax", quad_z(i, j, k + 1), c)

    faces = np.concatenate([int_faces] + bfaces, axis=0).astype(np.int32)
    owner = np.concatenate([int_own] + bown).astype(np.int32)
    F = len(faces)
    return PolyMesh(points=pts, faceOffsets=(np.arange(F + 1) * 4).astype(np.int32), facePoints=faces.ravel(),
                    owner=owner, neighbour=int_nei.astype(np.int32), patches=patches, nCells=nx * ny * nz)


def hex_subdomain(nLocal, grid, rank, lengths=None, jitter=0.0, seed=12345):
    """Sub-domain `rank` of a (grid[0]*nx, grid[1]*ny, grid[2]*nz)-cell block cut into grid boxes of
    nLocal=(nx,ny,nz) cells, generated directly (identical to decompose(hex_block(global),
    grid_partition) -- tests/test_decompose.py checks that), so an 8-rank 8M-cell case never needs the
    global mesh in one process.  Rank id = rx + ry*Px + rz*Px*Py."""
    from .decompose import SubDomain
    nx, ny, nz = nLocal
    Px, Py, Pz = grid
    rx, ry, rz = rank % Px, (rank // Px) % Py, rank // (Px * Py)
    GN = (nx * Px, ny * Py, nz * Pz)
    if lengths is None:
        lengths = (float(Px), float(Py), float(Pz))     # unit cube per sub-domain
    ox, oy, oz = rx * nx, ry * ny, rz * nz
    npx, npy = nx + 1, ny + 1

    def pid(i, j, k):
        return (i + j * npx + k * npx * npy).astype(np.int64)

    K, J, I = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    gi, gj, gk = I.ravel() + ox, J.ravel() + oy, K.ravel() + oz
    pts = _lattice_points(gi, gj, gk, GN, lengths, jitter, seed)
    gid = gi.astype(np.int64) + gj.astype(np.int64) * (GN[0] + 1) + gk.astype(np.int64) * (GN[0] + 1) * (GN[1] + 1)

    ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    cid = ci + cj * nx + ck * nx * ny

    def quad_x(i, j, k):
        return np.stack([pid(i, j, k), pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i, j, k + 1)], axis=1)

    def quad_y(i, j, k):
        return np.stack([pid(i, j, k), pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j, k)], axis=1)

    def quad_z(i, j, k):
        return np.stack([pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i, j + 1, k)], axis=1)

    faces3 = np.stack([quad_x(ci + 1, cj, ck), quad_y(ci, cj + 1, ck), quad_z(ci, cj, ck + 1)], axis=1)
    valid3 = np.stack([ci < nx - 1, cj < ny - 1, ck < nz - 1], axis=1)
    nei3 = np.stack([cid + 1, cid + nx, cid + nx * ny], axis=1)
    own3 = np.repeat(cid[:, None], 3, axis=1)
    int_faces, int_own, int_nei = faces3[valid3], own3[valid3], nei3[valid3]

    # the six sides: (name, cell mask, outward quad on the global-owner side, is max side, neighbour rank)
    def rk(ax, d):
        r3 = [rx, ry, rz]; r3[ax] += d
        return r3[0] + r3[1] * Px + r3[2] * Px * Py

    sides = [
        ("xmin", ci == 0, lambda i, j, k: quad_x(i, j, k), False, rx == 0, rk(0, -1)),
        ("xmax", ci == nx - 1, lambda i, j, k: quad_x(i + 1, j, k), True, rx == Px - 1, rk(0, +1)),
        ("ymin", cj == 0, lambda i, j, k: quad_y(i, j, k), False, ry == 0, rk(1, -1)),
        ("ymax", cj == ny - 1, lambda i, j, k: quad_y(i, j + 1, k), True, ry == Py - 1, rk(1, +1)),
        ("zmin", ck == 0, lambda i, j, k: quad_z(i, j, k), False, rz == 0, rk(2, -1)),
        ("zmax", ck == nz - 1, lambda i, j, k: quad_z(i, j, k + 1), True, rz == Pz - 1, rk(2, +1)),
    ]
    bfaces, bown, patches = [], [], []
    start = len(int_faces)
    for name, mask, quad, is_max, physical, _ in sides:
        if physical:
            q = quad(ci[mask], cj[mask], ck[mask])
            if not is_max:
                q = q[:, ::-1]            # same (arbitrary) outward ordering as hex_block's min patches
            bfaces.append(q); bown.append(cid[mask])
            patches.append(Patch(name, "patch", len(q), start)); start += len(q)
        else:
            patches.append(Patch(name, "patch", 0, start))
    procs = sorted([(nbr, name, mask, quad, is_max) for name, mask, quad, is_max, physical, nbr in sides if not physical],
                   key=lambda t: t[0])
    for nbr, name, mask, quad, is_max in procs:
        q = quad(ci[mask], cj[mask], ck[mask])
        if not is_max:
            q = q[:, [0, 3, 2, 1]]        # face::reverseFace of the global owner's face
        bfaces.append(q); bown.append(cid[mask])
        patches.append(Patch(f"procBoundary{rank}to{nbr}", "processor", len(q), start, myProcNo=rank, neighbProcNo=nbr))
        start += len(q)
    faces = np.concatenate([int_faces] + bfaces, axis=0).astype(np.int32)
    owner = np.concatenate([int_own] + bown).astype(np.int32)
    F = len(faces)
    mesh = PolyMesh(points=pts, faceOffsets=(np.arange(F + 1) * 4).astype(np.int32), facePoints=faces.ravel(), owner=owner,
                    neighbour=int_nei.astype(np.int32), patches=patches, nCells=nx * ny * nz)
    return SubDomain(mesh, rank, Px * Py * Pz, gid)


def read_obj_surface(path):
    """vertices (V,3) and polygon faces (list of 0-based index lists) of a Wavefront OBJ surface"""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                faces.append([int(tok.split("/")[0]) - 1 for tok in t[1:]])
    return np.array(verts, dtype=np.float64), faces


def extrude_surface(verts, faces2d, nLayers=15, thickness=1.5, direction=(0.0, 1.0, 0.0), box_patches=False) -> PolyMesh:
    """Linear extrusion of a planar polygon surface into prism/hex cells -- stands in for OpenFOAM's
    extrude2DMesh (reference testcase/run_serial:11, system/extrude2DMeshDict: nLayers 15, thickness 1.5 along
    +y; BASELINE.json configs[0]).  Numbering: point = v + layer*nV, cell = f + layer*nF; internal faces in
    upper-triangular order; patches back (layer 0), front (last layer), sides.
    box_patches=True mimics the testcase's topoSet + createPatch step (run_serial:12-13, system/createPatchDict):
    lateral faces on the bounding box of the surface go to side_left/right/bottom/top, back/front are called
    side_back/side_front, and what remains (the immersed shape) stays in defaultFaces -- the patch the testcase
    selects with -layerPatches '("def.*")'."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    nV, nF = len(verts), len(faces2d)
    # orient every polygon counter-clockwise about the extrusion direction
    polys = []
    for f in faces2d:
        p = verts[f]
        nrm = np.zeros(3)
        for i in range(len(f)):
            nrm += np.cross(p[i], p[(i + 1) % len(f)])
        polys.append(list(f) if nrm @ d > 0 else list(f)[::-1])
    pts = np.concatenate([verts + d * (thickness * l / nLayers) for l in range(nLayers + 1)], axis=0)
    # lateral adjacency of the 2-D faces
    edge_faces = {}
    for fi, f in enumerate(polys):
        for i in range(len(f)):
            a, b = f[i], f[(i + 1) % len(f)]
            edge_faces.setdefault((min(a, b), max(a, b)), []).append((fi, a, b))
    internal = []   # (owner, neighbour, vertex list)
    back, front, sides = [], [], []
    for l in range(nLayers):
        o0, o1 = l * nV, (l + 1) * nV
        for fi, f in enumerate(polys):
            c = fi + l * nF
            if l + 1 < nLayers:
                internal.append((c, c + nF, [v + o1 for v in f]))
            else:
                front.append((c, [v + o1 for v in f]))
            if l == 0:
                back.append((c, [v + o0 for v in f][::-1]))
            for i in range(len(f)):
                a, b = f[i], f[(i + 1) % len(f)]
                users = edge_faces[(min(a, b), max(a, b))]
                quad = [a + o0, b + o0, b + o1, a + o1]      # normal points out of this cell
                if len(users) == 1:
                    sides.append((c, quad))
                elif len(users) == 2:
                    other = users[0][0] if users[1][0] == fi else users[1][0]
                    if other > fi:
                        internal.append((c, other + l * nF, quad))
                else:
                    raise ValueError("non-manifold edge in the surface")
    internal.sort(key=lambda t: (t[0], t[1]))
    if box_patches:
        # two in-plane axes of the surface and its bounding box
        ax = [i for i in range(3) if abs(d[i]) < 0.5]
        lo, hi = verts[:, ax].min(axis=0), verts[:, ax].max(axis=0)
        tol = 1e-9 * float(np.max(hi - lo))
        groups = {k: [] for k in ("defaultFaces", "side_left", "side_right", "side_bottom", "side_top")}
        for c, quad in sides:
            q = pts[quad][:, ax]
            if np.all(np.abs(q[:, 0] - lo[0]) < tol): groups["side_left"].append((c, quad))
            elif np.all(np.abs(q[:, 0] - hi[0]) < tol): groups["side_right"].append((c, quad))
            elif np.all(np.abs(q[:, 1] - lo[1]) < tol): groups["side_bottom"].append((c, quad))
            elif np.all(np.abs(q[:, 1] - hi[1]) < tol): groups["side_top"].append((c, quad))
            else: groups["defaultFaces"].append((c, quad))
        named = [("defaultFaces", groups["defaultFaces"]), ("side_front", front), ("side_back", back)] + \
                [(k, groups[k]) for k in ("side_left", "side_right", "side_top", "side_bottom")]
    else:
        named = [("back", back), ("front", front), ("sides", sides)]
    named = [(k, g) for k, g in named if g]
    allf = [t[2] for t in internal] + [t[1] for _, g in named for t in g]
    owner = [t[0] for t in internal] + [t[0] for _, g in named for t in g]
    off = np.zeros(len(allf) + 1, np.int32)
    np.cumsum([len(f) for f in allf], out=off[1:])
    nI = len(internal)
    patches, startFace = [], nI
    for k, g in named:
        patches.append(Patch(k, "patch", len(g), startFace))
        startFace += len(g)
    return PolyMesh(points=pts, faceOffsets=off, facePoints=np.concatenate([np.asarray(f, np.int32) for f in allf]),
                    owner=np.asarray(owner, np.int32), neighbour=np.asarray([t[1] for t in internal], np.int32), patches=patches,
                    nCells=nF * nLayers)


def add_baffle(mesh: PolyMesh, select, name="baffle") -> PolyMesh:
    """createBaffles on a PolyMesh: the internal faces with select[f] true become PAIRS of boundary faces on the same points --
    the owner's face in patch `<name>_master`, the neighbour's (reversed, so that it points out of its cell) in `<name>_slave` --
    appended behind the existing patches; the remaining internal faces keep their order (the reference's testcase6 builds
    such a mesh with createBaffles, `testcase6/system/createBafflesDict`)."""
    nI = mesh.nInternalFaces
    select = np.asarray(select, bool)
    assert select.shape == (nI,) and select.any()
    off, fp = mesh.faceOffsets, mesh.facePoints

    def rows(ids, reverse=False):
        out = []
        for f in ids:
            r = fp[off[f]:off[f + 1]]
            out.append(np.concatenate([r[:1], r[:0:-1]]) if reverse else r)    # reversed face starts at the same point
        return out

    keep = np.flatnonzero(~select)
    gone = np.flatnonzero(select)
    old_b = np.arange(nI, mesh.nFaces)
    faces = rows(keep) + rows(old_b) + rows(gone) + rows(gone, reverse=True)
    owner = np.concatenate([mesh.owner[keep], mesh.owner[old_b], mesh.owner[gone], mesh.neighbour[gone]])
    patches = [Patch(name=p.name, type=p.type, nFaces=p.nFaces, startFace=p.startFace - len(gone)) for p in mesh.patches]
    start = mesh.nFaces - len(gone)
    patches.append(Patch(name=name + "_master", type="wall", nFaces=len(gone), startFace=start))
    patches.append(Patch(name=name + "_slave", type="wall", nFaces=len(gone), startFace=start + len(gone)))
    sizes = np.array([len(r) for r in faces], np.int64)
    return PolyMesh(points=mesh.points.copy(), faceOffsets=np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32),
                    facePoints=np.concatenate(faces).astype(np.int32), owner=owner.astype(np.int32),
                    neighbour=mesh.neighbour[keep].astype(np.int32), patches=patches, nCells=mesh.nCells)


def baffle_in_plane(lattice: PolyMesh, axis, value, also=None) -> np.ndarray:
    """selection for add_baffle: the internal faces of the UNJITTERED mesh `lattice` whose centre lies in the plane
    coordinate[axis] = value (and satisfies also(centres))"""
    nI = lattice.nInternalFaces
    off = lattice.faceOffsets
    n = off[1:nI + 1] - off[:nI]
    assert (n == n[0]).all()                   # (hex lattices: all quadrilaterals)
    ctr = lattice.points[lattice.facePoints[:off[nI]].reshape(nI, n[0])].mean(axis=1)
    sel = np.abs(ctr[:, axis] - value) < 1e-9
    return sel & also(ctr) if also is not None else sel


def split_baffles(mesh: PolyMesh, prefix="baffle") -> PolyMesh:
    """splitBaffles / mergeOrSplitBaffles -split (the reference's testcase6/run_serial:17-18) on the patches whose name starts
    with `prefix`: a point of those patches around which the cells fall into two groups not connected through an internal face
    that contains the point -- the interior of the wall, not its rim -- gets a twin with the same coordinates (appended behind
    the existing points); the second group's cells, and their faces, use the twin.  The wall becomes a slit of zero width."""
    off, fp = mesh.faceOffsets, mesh.facePoints.copy()
    nI = mesh.nInternalFaces
    cand = set()
    for p in mesh.patches:
        if p.name.startswith(prefix):
            cand |= set(fp[off[p.startFace]:off[p.startFace + p.nFaces]].tolist())
    faces_of = {}                                    # point -> faces that contain it
    face_of_pos = np.repeat(np.arange(mesh.nFaces), np.diff(off))
    for pos in np.flatnonzero(np.isin(fp, list(cand))):
        faces_of.setdefault(int(fp[pos]), []).append(int(face_of_pos[pos]))
    pts = [mesh.points]
    nP = mesh.nPoints
    for p in sorted(cand):
        cells = {}
        def find(c):
            while cells.setdefault(c, c) != c:
                cells[c] = cells[cells[c]]
                c = cells[c]
            return c
        for f in faces_of[p]:
            find(int(mesh.owner[f]))
            if f < nI:
                a, b = find(int(mesh.owner[f])), find(int(mesh.neighbour[f]))
                if a != b:
                    cells[a] = b
        roots = sorted({find(c) for c in list(cells)})
        if len(roots) < 2:
            continue
        assert len(roots) == 2
        keep = find(min(cells))                      # the group of the lowest cell keeps the point
        for f in faces_of[p]:
            if find(int(mesh.owner[f])) != keep:     # (an internal face lies inside one group: its owner decides)
                seg = fp[off[f]:off[f + 1]]
                seg[seg == p] = nP
        pts.append(mesh.points[p:p + 1])
        nP += 1
    return PolyMesh(points=np.concatenate(pts), faceOffsets=off.copy(), facePoints=fp, owner=mesh.owner.copy(), neighbour=mesh.neighbour.copy(),
                    patches=[Patch(name=q.name, type=q.type, nFaces=q.nFaces, startFace=q.startFace) for q in mesh.patches], nCells=mesh.nCells)
