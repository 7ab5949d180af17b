#!/usr/bin/env python3
"""Where the tail of the overlapped generation launch goes (k_generation, diagnostic build -DWA_GEN_TIME): wall-clock stamps of
mark block 0 relative to its own start, averaged over generations 0-19 and 200-499 of BASELINE config C3.

    python -m welding_robot_amd.build -DWA_GEN_TIME --out=$PWD/build/exp/gen_time.so
    WELDACS_LIB=$PWD/build/exp/gen_time.so python tools/gen_tail_time.py"""
import os
import sys

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
    solver.init_pheromone(1.0)
    solver.begin(p, ids[0], ids[1], streams=[0])
    def counters():
        out = np.zeros(16, np.uint64)
        ctx.check(ctx.lib.wa_acs_debug_counters(solver.h, out.ctypes.data, 1))
        return out

    counters()
    names = ("own ant done", "every ant done", "results in LDS", "ranked", "published + marked")
    for label, gens, show in (("generations 0-19", 20, True), ("20-199", 180, False), ("200-499", 300, True)):
        solver.run(gens)
        solver.sync()
        c = counters()
        if show and c[6]:
            print(label + ": " + ", ".join("%s %.2f us" % (nm, float(c[i]) / float(c[6]) / 100.0) for i, nm in enumerate(names)))


if __name__ == "__main__":
    main()
