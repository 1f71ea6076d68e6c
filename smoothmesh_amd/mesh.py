"""polyMesh container (what a constant/polyMesh directory holds) and point classification.

Mirrors the reference's setup either side of the loop: findInternalMeshPoints (src/smoothMesh.C:40-91)
and the features-off branch of classifyBoundaryPoints (src/boundaryPointSmoothing.C:301-340,397-420).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class Patch:
    name: str
    type: str          # patch | wall | processor | empty | ...
    nFaces: int
    startFace: int
    myProcNo: Optional[int] = None
    neighbProcNo: Optional[int] = None


@dataclass
class PolyMesh:
    points: np.ndarray        # (P, 3) f64
    faceOffsets: np.ndarray   # (F+1,) i32
    facePoints: np.ndarray    # (nnz,) i32
    owner: np.ndarray         # (F,) i32
    neighbour: np.ndarray     # (nInternalFaces,) i32
    patches: List[Patch] = field(default_factory=list)
    nCells: int = 0

    def __post_init__(self):
        self.points = np.ascontiguousarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.faceOffsets = np.ascontiguousarray(self.faceOffsets, dtype=np.int32)
        self.facePoints = np.ascontiguousarray(self.facePoints, dtype=np.int32)
        self.owner = np.ascontiguousarray(self.owner, dtype=np.int32)
        self.neighbour = np.ascontiguousarray(self.neighbour, dtype=np.int32)
        if not self.nCells:
            self.nCells = int(self.owner.max()) + 1 if len(self.owner) else 0

    @property
    def nPoints(self):
        return self.points.shape[0]

    @property
    def nFaces(self):
        return len(self.owner)

    @property
    def nInternalFaces(self):
        return len(self.neighbour)

    def find_internal_points(self) -> np.ndarray:
        """isInternalPoint, SM.C:40-91: true unless the point lies on a face of a non-processor patch;
        `empty` patches are refused like the reference does (SM.C:61-66)."""
        internal = np.ones(self.nPoints, dtype=np.uint8)
        for p in self.patches:
            if p.type == "processor":
                continue
            if p.type == "empty":
                raise ValueError("Smoothing of non-3D meshes (meshes with type empty patches) is not supported")
            a = self.faceOffsets[p.startFace]
            b = self.faceOffsets[p.startFace + p.nFaces]
            internal[self.facePoints[a:b]] = 0
        return internal

    def smoothing_surface_points(self) -> np.ndarray:
        """isSmoothingSurfacePoint with boundary point smoothing disabled (BPS.C:404-420): all false."""
        return np.zeros(self.nPoints, dtype=np.uint8)
