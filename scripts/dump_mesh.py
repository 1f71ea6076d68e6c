#!/usr/bin/env python3
"""dump a bench.py workload's mesh for scripts/native/setup_bench: python scripts/dump_mesh.py cavity215 /tmp/mesh.bin"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
kind, n, _ = bench.parse_workload(sys.argv[1])
m = bench.make_mesh(kind, n)
with open(sys.argv[2], "wb") as f:
    np.array([m.nPoints, m.nCells, m.nFaces, m.nInternalFaces], dtype=np.int32).tofile(f)
    np.ascontiguousarray(m.points, dtype=np.float64).tofile(f)
    np.ascontiguousarray(m.faceOffsets, dtype=np.int32).tofile(f)
    np.ascontiguousarray(m.facePoints, dtype=np.int32).tofile(f)
    np.ascontiguousarray(m.owner, dtype=np.int32).tofile(f)
    np.ascontiguousarray(m.neighbour, dtype=np.int32).tofile(f)
    np.ascontiguousarray(m.find_internal_points(), dtype=np.uint8).tofile(f)
