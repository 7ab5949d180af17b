#!/usr/bin/env python3
"""VERDICT r05 task 5 (the serial tail of a converged generation): BASELINE config C3 (128^3, 256 ants, 500 generations) with the dense sweep and with
lazy evaporation, by generation range -- wall clock of the UNTIMED loop (run(n) + sync) and, in a second pass, HIP events around every launch.
What a "dense while the ants explore, lazy once they have converged" hybrid could gain is the difference of the two in generations 200-499.

    python tools/converged_tail.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import api, synth  # noqa: E402

RANGES = (("0-19", 20), ("20-199", 180), ("200-499", 300))


def main():
    n, ants = 128, 256
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    p = api.default_params(max_iteration=500, predict=3.0 * n, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=1)
    best = {}
    for lazy in (False, True):
        solver = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants, lazy=lazy)
        name = "lazy " if lazy else "dense"
        for timed in (False, True):
            walls = {k: [] for k, _ in RANGES}
            for rep in range(4):
                solver.init_pheromone(1.0)
                solver.begin(p, ids[0], ids[1], streams=[0])
                ctx.sync()
                for label, gens in RANGES:
                    solver.profile(timed, 1)
                    t0 = time.perf_counter()
                    solver.run(gens)
                    solver.sync()
                    dt = time.perf_counter() - t0
                    if rep:
                        walls[label].append(dt / gens * 1e6)
                    if timed and rep == 3:
                        r = solver.profile_read()
                        print("%s events    %-8s " % (name, label) + "  ".join("%s %.2f us x %d" % (k, v["ms"] / max(1, v["launches"]) * 1e3, v["launches"]) for k, v in r.items() if v["launches"]), flush=True)
            if not timed:
                print("%s untimed   " % name + "   ".join("%s: %.2f us/generation (min of 3; %s)" % (k, min(v), " ".join("%.2f" % x for x in v)) for k, v in walls.items()), flush=True)
        c_, p_ = solver.results(1)
        best[lazy] = (float(c_[0]), p_[0])
        solver.close()
    print("same best cost and path:", best[False][0] == best[True][0] and np.array_equal(best[False][1], best[True][1]), best[False][0])


if __name__ == "__main__":
    main()
