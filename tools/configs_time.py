#!/usr/bin/env python3
"""BASELINE configs C1 and C2 on one GPU, end to end through the C ABI (STL file -> voxel grid on the device -> rank-based ACS), timed:
the small cases of BASELINE.json beside the C3 line of bench.py.   python tools/configs_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api

G = os.path.join(ROOT, "tests", "golden")
ctx = api.Context(0)
CASES = [("C1 files/cubic.stl, p 0.0219, wall 8, 16 ants, 50 generations", "cubic.stl", 0.0219, 8, (4, 4, 4), (20, 27, 20), 16, 50, 1.03),
         ("C2 files/simplified_piece.stl, p 0.0148, wall 4, 128 ants, 200 generations", "simplified_piece.stl", 0.0148, 4, None, None, 128, 200, 5.4126)]
for name, stl, prec, wall, a, b, ants, gens, predict in CASES:
    tris = api.stl_read_file(os.path.join(G, stl))
    ctx.sync()
    t0 = time.perf_counter()
    grid = api.Grid.from_mesh(ctx, tris, prec, wall)
    ctx.sync()
    t_vox = time.perf_counter() - t0
    cx, cy, cz = grid.coords()
    if a is None:   # the reference's own end points of this case (tests/golden: node ids 2177 and 48575)
        ids = np.array([2177, 48575], np.int64)
    else:
        ids = grid.resolve(np.array([[cx[a[2]], cy[a[1]], cz[a[0]]], [cx[b[2]], cy[b[1]], cz[b[0]]]], np.float32))
    s = api.AcsSolver(ctx, grid, 1, ants)
    p = api.default_params(max_iteration=gens, predict=predict, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=12345)
    s.solve(p, ids[0], ids[1]); s.init_pheromone(1.0)          # warm-up
    ctx.sync()
    t0 = time.perf_counter()
    s.solve(p, ids[0], ids[1])
    dt = time.perf_counter() - t0
    cost, path, _ = s.result()
    print("%s: grid %dx%dx%d (%d triangles) voxelised in %.2f ms; %d generations in %.2f ms = %.0f gen/s; best cost %.6f over %d nodes" % (
        name, grid.nx, grid.ny, grid.nz, len(tris), t_vox * 1e3, gens, dt * 1e3, gens / dt, cost, len(path)))
    s.close(); grid.close()
