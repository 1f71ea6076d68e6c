"""Domain decomposition in OpenFOAM decomposePar layout (processorN/constant/polyMesh).

The reference runs one MPI rank per sub-domain produced by decomposePar (testcase/run_parallel,
system/decomposeParDict); processor-patch points count as internal points (src/smoothMesh.C:49-58)
and per-point values are combined across ranks by syncTools::syncPointList.  decomposePar is not
available here, so this module writes the same layout:
  * local cells / points / faces keep ascending global order (cell-, point-, faceProcAddressing);
  * faces: internal, then every physical patch (kept even when empty), then one processor patch per
    neighbouring rank (ascending rank; faces in ascending global face id);
  * a processor face whose local cell is the global neighbour is reversed about its first vertex
    (face::reverseFace), so its normal points out of the local domain.
"""
from dataclasses import dataclass
from typing import List

import numpy as np

from .mesh import PolyMesh, Patch


@dataclass
class SubDomain:
    mesh: PolyMesh
    rank: int
    nRanks: int
    pointProcAddressing: np.ndarray   # local point -> global point id
    cellProcAddressing: np.ndarray = None
    faceProcAddressing: np.ndarray = None

    def processor_patch_points(self) -> np.ndarray:
        """sorted unique GLOBAL ids of the points on this rank's processor patches"""
        m = self.mesh
        ids = []
        for p in m.patches:
            if p.type == "processor" and p.nFaces:
                a, b = m.faceOffsets[p.startFace], m.faceOffsets[p.startFace + p.nFaces]
                ids.append(m.facePoints[a:b])
        if not ids:
            return np.zeros(0, np.int64)
        return np.unique(self.pointProcAddressing[np.concatenate(ids)])


    def processor_patch_point_lists(self) -> dict:
        """{neighbour rank: sorted unique GLOBAL ids of the points on this rank's processor patch to that neighbour}"""
        m = self.mesh
        out = {}
        for p in m.patches:
            if p.type == "processor" and p.nFaces:
                a, b = m.faceOffsets[p.startFace], m.faceOffsets[p.startFace + p.nFaces]
                ids = np.unique(self.pointProcAddressing[m.facePoints[a:b]])
                o = int(p.neighbProcNo)
                out[o] = np.union1d(out[o], ids) if o in out else ids
        return out


def _connected_components(n, ea, eb):
    """labels 0 .. k-1 of the connected components of the undirected graph on n nodes with edges (ea[i], eb[i]), numbered in the
    order of their smallest node (the numbering scipy.sparse.csgraph.connected_components gives; numpy only: minimum-label
    propagation with pointer jumping -- the graphs here are chains of a few copies of a point, a handful of rounds)"""
    lab = np.arange(n, dtype=np.int64)
    ea = np.asarray(ea, np.int64); eb = np.asarray(eb, np.int64)
    while len(ea):
        m = np.minimum(lab[ea], lab[eb])
        new = lab.copy()
        np.minimum.at(new, ea, m)
        np.minimum.at(new, eb, m)
        new = new[new]
        if np.array_equal(new, lab):
            break
        lab = new
    return np.unique(lab, return_inverse=True)[1].astype(np.int64)


def shared_point_components(patch_lists):
    """The copies of points that syncTools::syncPointList combines, as OpenFOAM's globalPoints finds them: the copies of a point
    on the two sides of a PROCESSOR PATCH are the same point, and so is everything connected through such pairs (transitive
    closure; a rank has one local point per mesh point, which joins all its patches) -- and nothing else.  Two ranks that hold
    the same mesh point but are connected by no chain of processor faces through it do NOT share it: the two sides of a baffle
    (createBaffles; the reference's testcase6) on different ranks, or two cells that touch in a point only across a solid
    region.  On a manifold mesh this is "every rank that holds the point on a processor patch".
    patch_lists[r] = {neighbour: sorted global ids on r's patch to it} (SubDomain.processor_patch_point_lists).
    -> (node_rank, node_gid, comp, comp_size): one node per (rank, point on one of its processor patches), its component label
    and that component's number of members."""
    nR = len(patch_lists)
    ids = [np.unique(np.concatenate([np.asarray(v, np.int64) for v in lists.values()])) if lists else np.zeros(0, np.int64) for lists in patch_lists]
    base = np.concatenate([[0], np.cumsum([len(i) for i in ids])]).astype(np.int64)
    n = int(base[-1])
    node_rank = np.concatenate([np.full(len(i), r, np.int32) for r, i in enumerate(ids)]) if n else np.zeros(0, np.int32)
    node_gid = np.concatenate(ids) if n else np.zeros(0, np.int64)
    ea, eb = [], []
    for r, lists in enumerate(patch_lists):
        for o, mine in lists.items():
            o = int(o)
            if o <= r or r not in patch_lists[o]:
                continue
            common = np.intersect1d(np.asarray(mine, np.int64), np.asarray(patch_lists[o][r], np.int64))
            ea.append(base[r] + np.searchsorted(ids[r], common))
            eb.append(base[o] + np.searchsorted(ids[o], common))
    if n == 0:
        return node_rank, node_gid, np.zeros(0, np.int32), np.zeros(0, np.int64)
    ea = np.concatenate(ea) if ea else np.zeros(0, np.int64)
    eb = np.concatenate(eb) if eb else np.zeros(0, np.int64)
    comp = _connected_components(n, ea, eb)
    return node_rank, node_gid, comp.astype(np.int64), np.bincount(comp)[comp]


def shared_point_groups(patch_lists):
    """-> {global id: [sorted rank lists, one per group with >= 2 members]} (shared_point_components, for tests and small cases)"""
    node_rank, node_gid, comp, size = shared_point_components(patch_lists)
    groups = {}
    by_comp = {}
    for r, g, c, k in zip(node_rank.tolist(), node_gid.tolist(), comp.tolist(), size.tolist()):
        if k >= 2:
            by_comp.setdefault((g, c), []).append(r)
    for (g, _), ranks in by_comp.items():
        groups.setdefault(g, []).append(sorted(ranks))
    return groups


def grid_partition(mesh: PolyMesh, grid) -> np.ndarray:
    """decomposePar 'simple'-like geometric partition: split the bounding box of the cell-centre
    estimates (mean of the cell's face vertex averages) into grid[0] x grid[1] x grid[2] boxes."""
    px, py, pz = grid
    F = mesh.nFaces
    sizes = np.diff(mesh.faceOffsets)
    fsum = np.zeros((F, 3))
    np.add.at(fsum, np.repeat(np.arange(F), sizes), mesh.points[mesh.facePoints])
    fc = fsum / sizes[:, None]
    cs = np.zeros((mesh.nCells, 3)); cn = np.zeros(mesh.nCells)
    np.add.at(cs, mesh.owner, fc); np.add.at(cn, mesh.owner, 1)
    np.add.at(cs, mesh.neighbour, fc[:mesh.nInternalFaces]); np.add.at(cn, mesh.neighbour, 1)
    cc = cs / cn[:, None]
    lo, hi = mesh.points.min(0), mesh.points.max(0)
    rel = (cc - lo) / (hi - lo)
    r = [np.minimum((rel[:, a] * g).astype(np.int64), g - 1) for a, g in enumerate((px, py, pz))]
    return (r[0] + r[1] * px + r[2] * px * py).astype(np.int32)


def _take_faces(mesh, fids, reverse):
    """CSR sub-list of faces `fids`; faces with reverse[i] get reverseFace ordering"""
    off, fp = mesh.faceOffsets, mesh.facePoints
    b = off[fids].astype(np.int64); e = off[fids + 1].astype(np.int64)
    n = e - b
    newoff = np.zeros(len(fids) + 1, np.int64); np.cumsum(n, out=newoff[1:])
    j = np.arange(newoff[-1]) - np.repeat(newoff[:-1], n)
    bb = np.repeat(b, n); ee = np.repeat(e, n); rv = np.repeat(reverse, n)
    src = np.where(rv & (j > 0), ee - j, bb + j)
    return newoff, fp[src]


def decompose(mesh: PolyMesh, cellRank: np.ndarray, nRanks: int) -> List[SubDomain]:
    cellRank = np.asarray(cellRank)
    nIF = mesh.nInternalFaces
    own_r = cellRank[mesh.owner]
    nei_r = cellRank[mesh.neighbour]
    subs = []
    for r in range(nRanks):
        cells = np.flatnonzero(cellRank == r)
        g2l_cell = np.full(mesh.nCells, -1, np.int64); g2l_cell[cells] = np.arange(len(cells))
        intf = np.flatnonzero((own_r[:nIF] == r) & (nei_r == r))
        groups = [(intf, np.zeros(len(intf), bool))]
        own_l = [g2l_cell[mesh.owner[intf]]]
        nei_l = g2l_cell[mesh.neighbour[intf]]
        patches = []
        start = len(intf)
        for p in mesh.patches:
            f = np.arange(p.startFace, p.startFace + p.nFaces)
            f = f[own_r[f] == r]
            groups.append((f, np.zeros(len(f), bool)))
            own_l.append(g2l_cell[mesh.owner[f]])
            patches.append(Patch(p.name, p.type, len(f), start))
            start += len(f)
        # processor faces
        cut = np.flatnonzero(((own_r[:nIF] == r) | (nei_r == r)) & (own_r[:nIF] != nei_r))
        mine_is_owner = own_r[cut] == r
        other = np.where(mine_is_owner, nei_r[cut], own_r[cut])
        for o in np.unique(other):
            sel = other == o
            f = cut[sel]
            rev = ~mine_is_owner[sel]
            groups.append((f, rev))
            own_l.append(g2l_cell[np.where(rev, mesh.neighbour[f], mesh.owner[f])])
            patches.append(Patch(f"procBoundary{r}to{int(o)}", "processor", len(f), start, myProcNo=r, neighbProcNo=int(o)))
            start += len(f)
        fids = np.concatenate([g[0] for g in groups])
        rev = np.concatenate([g[1] for g in groups])
        off, fpg = _take_faces(mesh, fids, rev)
        gpts = np.unique(fpg)
        fpl = np.searchsorted(gpts, fpg)
        sub = PolyMesh(points=mesh.points[gpts], faceOffsets=off.astype(np.int32), facePoints=fpl.astype(np.int32),
                       owner=np.concatenate(own_l).astype(np.int32), neighbour=nei_l.astype(np.int32), patches=patches,
                       nCells=len(cells))
        subs.append(SubDomain(sub, r, nRanks, gpts.astype(np.int64), cells.astype(np.int64), fids.astype(np.int64)))
    return subs


def shared_point_table(subs: List[SubDomain]):
    """Global view (tests / single-process drivers): CSR over the shared points (shared_point_components: copies connected
    through processor patches) -> (offsets, domain ids ascending, local ids).  Ordered by global id, then by the lowest sharer."""
    node_rank, node_gid, comp, size = shared_point_components([s.processor_patch_point_lists() for s in subs])
    keep = size >= 2
    node_rank, node_gid, comp = node_rank[keep], node_gid[keep], comp[keep]
    if len(comp) == 0:
        return np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    low = np.full(int(comp.max()) + 1, len(subs), np.int64)
    np.minimum.at(low, comp, node_rank)
    order = np.lexsort((node_rank, comp, low[comp], node_gid))
    node_rank, node_gid, comp = node_rank[order], node_gid[order], comp[order]
    first = np.concatenate([[True], (comp[1:] != comp[:-1])])
    off = np.concatenate([np.flatnonzero(first), [len(comp)]]).astype(np.int32)
    loc = np.zeros(len(comp), np.int32)
    for s in subs:
        m = node_rank == s.rank
        o = np.argsort(s.pointProcAddressing, kind="stable")
        loc[m] = o[np.searchsorted(s.pointProcAddressing[o], node_gid[m])]
    return off, node_rank.astype(np.int32), loc


def bfs_partition(mesh: PolyMesh, nRanks: int, seed: int = 0, island: bool = False) -> np.ndarray:
    """Irregular partition, standing in for decomposePar's scotch method (testcase/system/decomposeParDict:9-11), which
    is not available here: nRanks regions grown breadth-first over the cell-cell graph from random seed cells.  The
    interfaces are ragged (no planes), sub-domain sizes differ, points are shared by 2..6 ranks off any lattice pattern.
    island=True additionally hands a small blob of cells deep inside another region to rank 0, so that one sub-domain
    is DISCONNECTED (scotch does produce those) and one rank sits inside another."""
    rng = np.random.default_rng(seed)
    C, nIF = mesh.nCells, mesh.nInternalFaces
    own, nei = mesh.owner[:nIF].astype(np.int64), mesh.neighbour.astype(np.int64)
    rank = np.full(C, -1, np.int32)
    seeds = rng.choice(C, size=nRanks, replace=False)
    rank[seeds] = np.arange(nRanks, dtype=np.int32)
    order = rng.permutation(nIF)                      # which neighbour claims a cell when several could: random, reproducible
    o, n = own[order], nei[order]
    while (rank < 0).any():
        before = int((rank < 0).sum())
        ro, rn = rank[o], rank[n]
        a = (ro >= 0) & (rn < 0)
        b = (rn >= 0) & (ro < 0)
        rank[n[a]] = ro[a]
        rank[o[b]] = rn[b]
        if int((rank < 0).sum()) == before:           # a mesh of several components: seed the next one
            left = np.flatnonzero(rank < 0)
            rank[left[0]] = int(rng.integers(nRanks))
    if island and nRanks > 1:
        # cells all of whose face neighbours are of the same (non-zero) rank: deep inside it; take one and its neighbours
        same = np.ones(C, bool)
        np.logical_and.at(same, own, rank[own] == rank[nei])
        np.logical_and.at(same, nei, rank[own] == rank[nei])
        cand = np.flatnonzero(same & (rank != 0))
        if len(cand):
            c = int(cand[rng.integers(len(cand))])
            blob = np.concatenate([[c], nei[own == c], own[nei == c]])
            rank[blob] = 0
    return rank


def random_partition(mesh: PolyMesh, nRanks: int, seed: int = 0) -> np.ndarray:
    """every cell to a random rank: the worst case for the shared-point tables (nearly every point is shared, by up to
    eight ranks; sub-domains are clouds of cells touching in faces, edges and single points)"""
    rng = np.random.default_rng(seed)
    r = rng.integers(nRanks, size=mesh.nCells).astype(np.int32)
    r[rng.choice(mesh.nCells, size=nRanks, replace=False)] = np.arange(nRanks, dtype=np.int32)   # no empty rank
    return r


def merge_disjoint(a: PolyMesh, b: PolyMesh, offset=(0.0, 0.0, 0.0), suffix="_b") -> PolyMesh:
    """two meshes that do not touch as ONE polyMesh (b's points moved by offset, its patches renamed): a rank that holds all of b
    then has no shared point at all"""
    nIa, nIb = a.nInternalFaces, b.nInternalFaces
    Pa, Ca = a.nPoints, a.nCells

    def faces(m, lo, hi, shift):
        off = m.faceOffsets.astype(np.int64)
        return np.diff(off[lo:hi + 1]), m.facePoints[off[lo]:off[hi]].astype(np.int64) + shift

    parts = [faces(a, 0, nIa, 0), faces(b, 0, nIb, Pa), faces(a, nIa, a.nFaces, 0), faces(b, nIb, b.nFaces, Pa)]
    sizes = np.concatenate([p[0] for p in parts])
    off = np.zeros(len(sizes) + 1, np.int64); np.cumsum(sizes, out=off[1:])
    owner = np.concatenate([a.owner[:nIa], b.owner[:nIb] + Ca, a.owner[nIa:], b.owner[nIb:] + Ca])
    neighbour = np.concatenate([a.neighbour, b.neighbour + Ca])
    patches = []
    for p in a.patches:
        patches.append(Patch(p.name, p.type, p.nFaces, p.startFace + nIb))
    nBa = a.nFaces - nIa
    for p in b.patches:
        patches.append(Patch(p.name + suffix, p.type, p.nFaces, p.startFace - nIb + nIa + nIb + nBa))
    pts = np.concatenate([a.points.reshape(-1, 3), b.points.reshape(-1, 3) + np.asarray(offset, np.float64)])
    return PolyMesh(points=pts, faceOffsets=off.astype(np.int32), facePoints=np.concatenate([p[1] for p in parts]).astype(np.int32),
                    owner=owner.astype(np.int32), neighbour=neighbour.astype(np.int32), patches=patches, nCells=Ca + b.nCells)
