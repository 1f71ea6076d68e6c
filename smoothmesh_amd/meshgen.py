"""Synthetic polyMesh generators standing in for OpenFOAM's blockMesh (unavailable here).

hex_block() reproduces blockMesh's single-block numbering: point id
i + j(nx+1) + k(nx+1)(ny+1), cell id i + j nx + k nx ny, internal faces in upper-triangular order
(per cell: +x, +y, +z neighbour), then the six boundary patches.  Interior points can be jittered
with a seeded PRNG so that smoothing has work to do (a uniform block is a fixed point,
src/smoothMesh.C:2401).
"""
import numpy as np

from .mesh import PolyMesh, Patch


def hex_block(nx, ny=None, nz=None, lengths=(1.0, 1.0, 1.0), jitter=0.0, seed=12345) -> PolyMesh:
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    npx, npy = nx + 1, ny + 1
    hx, hy, hz = lengths[0] / nx, lengths[1] / ny, lengths[2] / nz

    def pid(i, j, k):
        return (i + j * npx + k * npx * npy).astype(np.int64)

    K, J, I = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    pts = np.stack([I.ravel() * hx, J.ravel() * hy, K.ravel() * hz], axis=1).astype(np.float64)
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        interior = ((I > 0) & (I < nx) & (J > 0) & (J < ny) & (K > 0) & (K < nz)).ravel()
        d = rng.uniform(-jitter, jitter, size=(pts.shape[0], 3)) * np.array([hx, hy, hz])
        pts[interior] += d[interior]

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
