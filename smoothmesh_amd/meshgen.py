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
