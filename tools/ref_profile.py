#!/usr/bin/env python3
"""REF mode at C3: wall time by generation range (chunks of 20 generations, synchronised), to see where a 500-generation REF run spends
its time -- the exploring generations walk ant after ant; once converged the generation is speculated (k_ref_draws / k_walk_ref_spec)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from welding_robot_amd import api, synth  # noqa: E402

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ctx = api.Context(0)
n = 128
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=256)
p = api.default_params(max_iteration=gens, predict=731.43, fixed_colony=256, rng_mode=api.RNG_REF)
s.srand(12345)
s.init_pheromone(1.0)
s.begin(p, ids[0], ids[1])
done = 0
while done < gens:
    c = min(20, gens - done)
    t0 = time.perf_counter()
    s.run(c)
    s.sync()
    dt = time.perf_counter() - t0
    tr = s.trace()
    print("generations %3d..%3d: %7.2f ms per generation, %6.0f steps per generation, finite %d" % (done, done + c - 1, dt * 1e3 / c, tr["steps"][done:done + c].mean(), int(tr["finite"][done + c - 1])))
    done += c
