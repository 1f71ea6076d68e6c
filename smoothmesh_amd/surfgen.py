"""Geometry inputs of the boundary point smoothing: Wavefront OBJ readers / writers for the target surface and the
feature edge meshes (constant/geometry/{targetSurfaces,initEdges,targetEdges}.obj, SM.C:1924-1926) and small synthetic
generators for tests and examples.

Reading follows what OpenFOAM's readers (third-party, not in the reference tree) do with these files: polygons of a
surface are triangulated as a fan about their first vertex (OBJsurfaceFormat with a triangle face type), an edge mesh
takes the consecutive pairs of every `l` record and drops the points no edge uses, keeping the order of the rest
(OBJedgeFormat)."""
import numpy as np


def _vertex_index(tok, nv):
    i = int(tok.split("/")[0])
    return i - 1 if i > 0 else nv + i


def read_obj_surface(path):
    """-> (points (n,3) f64, triangles (m,3) i32)"""
    pts, tris = [], []
    with open(path) as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "v":
                pts.append([float(w[1]), float(w[2]), float(w[3])])
            elif w[0] == "f":
                v = [_vertex_index(t, len(pts)) for t in w[1:]]
                for k in range(1, len(v) - 1):
                    tris.append([v[0], v[k], v[k + 1]])
    return np.array(pts, np.float64).reshape(-1, 3), np.array(tris, np.int32).reshape(-1, 3)


def read_obj_edges(path):
    """-> (points (n,3) f64, edges (m,2) i32); unused points dropped"""
    pts, edges = [], []
    with open(path) as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "v":
                pts.append([float(w[1]), float(w[2]), float(w[3])])
            elif w[0] == "l":
                v = [_vertex_index(t, len(pts)) for t in w[1:]]
                for k in range(len(v) - 1):
                    edges.append([v[k], v[k + 1]])
    pts = np.array(pts, np.float64).reshape(-1, 3)
    edges = np.array(edges, np.int32).reshape(-1, 2)
    used = np.zeros(len(pts), bool)
    used[edges.ravel()] = True
    if not used.all():
        remap = np.cumsum(used) - 1
        pts, edges = pts[used], remap[edges].astype(np.int32)
    return pts, edges


def write_obj_surface(path, pts, tris):
    with open(path, "w") as f:
        f.write("# target surface\n")
        for p in np.asarray(pts):
            f.write("v %.17g %.17g %.17g\n" % tuple(p))
        for t in np.asarray(tris):
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def write_obj_edges(path, pts, edges):
    with open(path, "w") as f:
        f.write("# feature edges\n")
        for p in np.asarray(pts):
            f.write("v %.17g %.17g %.17g\n" % tuple(p))
        for e in np.asarray(edges):
            f.write("l %d %d\n" % (e[0] + 1, e[1] + 1))


def _lattice_to_xyz(ijk, m, lo, hi):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    return lo + (hi - lo) * (np.asarray(ijk, np.float64) / m)


def box_surface(m, lo=(0, 0, 0), hi=(1, 1, 1), warp=None):
    """Triangulated surface of a box, m x m quads (two triangles each) per side, points shared along the box edges.
    warp: optional map points (n,3) -> points (n,3) applied to the result."""
    ids, pts, tris = {}, [], []

    def pid(i, j, k):
        key = (i, j, k)
        if key not in ids:
            ids[key] = len(pts)
            pts.append(key)
        return ids[key]

    for axis in range(3):
        for side in (0, m):
            for a in range(m):
                for b in range(m):
                    def node(da, db):
                        c = [0, 0, 0]
                        c[axis] = side
                        c[(axis + 1) % 3] = a + da
                        c[(axis + 2) % 3] = b + db
                        return pid(*c)
                    q = [node(0, 0), node(1, 0), node(1, 1), node(0, 1)]
                    if side == 0:
                        q = q[::-1]
                    tris.append([q[0], q[1], q[2]])
                    tris.append([q[0], q[2], q[3]])
    xyz = _lattice_to_xyz(np.array(pts), m, lo, hi)
    if warp is not None:
        xyz = warp(xyz)
    return np.ascontiguousarray(xyz, np.float64), np.array(tris, np.int32)


def box_feature_edges(m, lo=(0, 0, 0), hi=(1, 1, 1), warp=None):
    """The twelve edges of a box as an edge mesh, m segments each, sharing the eight corner points."""
    ids, pts, edges = {}, [], []

    def pid(key):
        if key not in ids:
            ids[key] = len(pts)
            pts.append(key)
        return ids[key]

    for axis in range(3):
        for u in (0, m):
            for v in (0, m):
                for a in range(m):
                    c0, c1 = [0, 0, 0], [0, 0, 0]
                    c0[axis], c1[axis] = a, a + 1
                    c0[(axis + 1) % 3] = c1[(axis + 1) % 3] = u
                    c0[(axis + 2) % 3] = c1[(axis + 2) % 3] = v
                    edges.append([pid(tuple(c0)), pid(tuple(c1))])
    xyz = _lattice_to_xyz(np.array(pts), m, lo, hi)
    if warp is not None:
        xyz = warp(xyz)
    return np.ascontiguousarray(xyz, np.float64), np.array(edges, np.int32)


def sphere_surface(centre=(0.5, 0.5, 0.5), radius=0.25, levels=3):
    """Triangulated sphere: an octahedron subdivided `levels` times (4^levels * 8 triangles), vertices on the sphere."""
    pts = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    tris = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    pts = [np.array(p, np.float64) for p in pts]
    for _ in range(levels):
        mid, out = {}, []

        def midpoint(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                m = pts[a] + pts[b]
                pts.append(m / np.linalg.norm(m))
                mid[key] = len(pts) - 1
            return mid[key]

        for a, b, c in tris:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            out += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        tris = out
    xyz = np.asarray(centre, np.float64) + radius * np.array(pts)
    return np.ascontiguousarray(xyz), np.array(tris, np.int32)
