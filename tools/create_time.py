#!/usr/bin/env python3
"""Time solver creation (device allocation + field initialisation) for the dense and the lazy solver."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
for lazy in (False, True, False, True):
    for slots in (8, 32):
        ctx.sync(); t0 = time.perf_counter()
        s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=lazy)
        ctx.sync(); t1 = time.perf_counter()
        s.close(); ctx.sync(); t2 = time.perf_counter()
        print("lazy=%s slots=%d create %.3f s, destroy %.3f s" % (lazy, slots, t1 - t0, t2 - t1))
