#!/usr/bin/env python3
"""Per-launch times of the generation loop (HIP events around every launch of every generation, so the loop itself is slower than
untimed): BASELINE config C3, generations 0-19 (exploratory) and 200-499 (converged).

    python tools/gen_loop_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import api, synth  # noqa: E402


def main():
    n, ants = 128, 256
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    solver = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants)
    p = api.default_params(max_iteration=500, predict=3.0 * n, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=1)
    for rep in range(2):
        solver.init_pheromone(1.0)
        solver.begin(p, ids[0], ids[1], streams=[0])
        ctx.sync()
        for label, gens, timed in (("generations 0-19", 20, True), ("20-199", 180, False), ("200-499", 300, True)):
            solver.profile(timed, 1)
            t0 = time.perf_counter()
            solver.run(gens)
            solver.sync()
            dt = time.perf_counter() - t0
            if timed and rep == 1:
                r = solver.profile_read()
                print("%-18s wall %7.1f us/generation   " % (label, dt / gens * 1e6) +
                      "  ".join("%s %.1f us x %d" % (k, v["ms"] / max(1, v["launches"]) * 1e3, v["launches"]) for k, v in r.items() if v["launches"]))
    print("best:", solver.results(1)[0])


if __name__ == "__main__":
    main()
