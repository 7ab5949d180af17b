#!/usr/bin/env python3
"""Time the voxelisation (mesh -> occupancy on the device) at several precisions: the triangle-clipped kernel
(default) and the O(T*N^3) form (WA_VOXELIZE_DENSE=1); check against the oracle at the small one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api
ctx = api.Context(0)
MESH = sys.argv[1] if len(sys.argv) > 1 else "simplified_piece.stl"   # or origin_piece.stl: the reference's largest mesh (29 888 triangles)
tris = api.stl_read_file(os.path.join(ROOT, "tests", "golden", MESH))
def timed(p, wall, dense):
    if dense:
        os.environ["WA_VOXELIZE_DENSE"] = "1"
    try:
        api.Grid.from_mesh(ctx, tris, p, wall).close()
        t0 = time.perf_counter(); g = api.Grid.from_mesh(ctx, tris, p, wall); ctx.sync(); t1 = time.perf_counter()
    finally:
        os.environ.pop("WA_VOXELIZE_DENSE", None)
    return g, t1 - t0


for p, wall in [(0.0148, 4), (0.0065, 4), (0.0033, 4), (0.0021, 4), (0.0012, 4)]:
    g, t = timed(p, wall, False)
    gd, td = timed(p, wall, True)
    tests = g.n * len(tris)
    same = bool(np.array_equal(g.occupancy(), gd.occupancy()))
    gd.close()
    print("p=%.4f grid %dx%dx%d = %d voxels x %d tris: clipped %.2f ms, dense form %.1f ms (%.2e plane tests, %.0f Gtests/s), equal: %s, free %d" % (
        p, g.nx, g.ny, g.nz, g.n, len(tris), t * 1e3, td * 1e3, tests, tests / td / 1e9, same, g.n_free))
    if p == 0.0148:
        import oracle_lib as O
        og = O.grid_from_mesh(O.stl_parse(open(os.path.join(ROOT, "tests", "golden", MESH), "rb").read()), p, wall)
        print("   equals oracle:", bool(np.array_equal(g.occupancy(), og.free)))
    g.close()
