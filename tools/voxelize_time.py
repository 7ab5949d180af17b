#!/usr/bin/env python3
"""Time k_voxelize (mesh -> occupancy on the device) at several precisions; check against the oracle at the small one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api
ctx = api.Context(0)
tris = api.stl_read_file(os.path.join(ROOT, "tests", "golden", "simplified_piece.stl"))
for p, wall in [(0.0148, 4), (0.0065, 4), (0.0033, 4), (0.0021, 4)]:
    api.Grid.from_mesh(ctx, tris, p, wall).close()
    t0 = time.perf_counter(); g = api.Grid.from_mesh(ctx, tris, p, wall); ctx.sync(); t1 = time.perf_counter()
    tests = g.n * len(tris)
    print("p=%.4f grid %dx%dx%d = %d voxels x %d tris = %.2e plane tests: %.1f ms (%.1f Gtests/s), free %d" % (
        p, g.nx, g.ny, g.nz, g.n, len(tris), tests, (t1 - t0) * 1e3, tests / (t1 - t0) / 1e9, g.n_free))
    if p == 0.0148:
        import oracle_lib as O
        og = O.grid_from_mesh(O.stl_parse(open(os.path.join(ROOT, "tests", "golden", "simplified_piece.stl"), "rb").read()), p, wall)
        print("   equals oracle:", bool(np.array_equal(g.occupancy(), og.free)))
    g.close()
