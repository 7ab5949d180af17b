#!/usr/bin/env python3
"""Per-kernel split of one C5-shaped batch (256^3, 48 concurrent pair searches, 24 ants, 150 generations), lazy solver."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
n, slots, gens = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 48, 150
free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, 64, seed=7)
pairs = [(i, j) for i in range(64) for j in range(i + 1, 64)][:slots]
s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
p = api.default_params(max_iteration=gens, predict=24 / 0.35, rng_mode=api.RNG_DEV, seed=7)
a, b = [int(pts[i]) for i, _ in pairs], [int(pts[j]) for _, j in pairs]
s.solve(p, a, b, streams=list(range(slots))); s.reset_pheromone(1.0)
ctx.sync(); t0 = time.perf_counter(); s.solve(p, a, b, streams=list(range(slots))); t = time.perf_counter() - t0
s.reset_pheromone(1.0); s.profile(True, 1); s.solve(p, a, b, streams=list(range(slots)))
pr = s.profile_read()
tr = s.trace()
print("%d^3, %d slots x 24 ants, %d generations: %.1f ms = %.0f us per batch-generation (%.0f pair-generations/s)" % (
    n, slots, gens, t * 1e3, t / gens * 1e6, slots * gens / t))
print("  per-dispatch averages: " + ", ".join("%s %.1f us" % (k, v["ms"] / v["launches"] * 1e3) for k, v in pr.items() if v["launches"]))
print("  slot 0: steps/ant first/last generation %.0f / %.0f" % (tr["steps"][0] / 24, tr["steps"][-1] / max(tr["colony"][-1], 1)))
