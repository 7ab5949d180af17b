#!/usr/bin/env python3
"""Where a C5 run spends its time outside the kernels of a generation (diagnostic): begin / run / reset / results per batch."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
n, P = 256, 64
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 224
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, P, seed=7)
pairs = sorted([(i, j) for i in range(P) for j in range(i + 1, P)], key=lambda t: (t[1], t[0]))
t0 = time.perf_counter()
s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
ctx.sync()
print("create %.3f s" % (time.perf_counter() - t0))
p = api.default_params(max_iteration=150, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
for b0 in range(0, len(pairs), slots):
    idx = pairs[b0:b0 + slots]
    t0 = time.perf_counter(); s.begin(p, [pts[a] for a, b in idx], [pts[b] for a, b in idx], streams=list(range(b0, b0 + len(idx)))); ctx.sync(); t1 = time.perf_counter()
    s.run(150); s.sync(); t2 = time.perf_counter()
    s.reset_pheromone(1.0); ctx.sync(); t3 = time.perf_counter()
    s.results(len(idx)); t4 = time.perf_counter()
    print("batch at %4d: begin %.4f run %.4f reset %.4f results %.4f" % (b0, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
