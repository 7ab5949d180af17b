#!/usr/bin/env python3
"""Lazy evaporation (wa_acs_create_lazy) beside the dense sweep on the benchmark search (128^3, 256 ants, 500 generations,
per-dispatch profiling off): generations/s, and that trace, path and the full pheromone field are identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api, synth

ctx = api.Context(0)
for n, ants, gens in ((128, 256, 500), (256, 256, 300)):
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    p = api.default_params(max_iteration=gens, predict=731.43 * n / 128, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=12345)
    ref = None
    for lazy in (False, True):
        s = api.AcsSolver(ctx, grid, 1, ants, lazy=lazy)
        s.solve(p, ids[0], ids[1]); s.reset_pheromone(1.0)
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.perf_counter(); s.solve(p, ids[0], ids[1]); best = min(best, time.perf_counter() - t0)
            if rep < 2:
                s.reset_pheromone(1.0)
        s.reset_pheromone(1.0); s.profile(True, 1); s.solve(p, ids[0], ids[1])
        pr = s.profile_read()
        per = ", ".join("%s %.1f us" % (k, v["ms"] / v["launches"] * 1e3) for k, v in pr.items() if v["launches"])
        s.profile(False, 1)
        tr, (cost, path, _) = s.trace(), s.result()
        key = (tr["bestL"].tobytes(), tr["steps"].tobytes(), path.tobytes(), s.pheromone().tobytes())
        ref = ref or key
        print("%d^3, %d ants, %d generations, %s: %.2f ms = %.0f gen/s (%.1f us/generation); cost %.1f; identical to dense: %s" % (
            n, ants, gens, "lazy evaporation" if lazy else "dense sweep     ", best * 1e3, gens / best, best / gens * 1e6, cost, key == ref))
        print("      per-dispatch averages (profiled rerun): " + per)
        s.close()
    grid.close()
