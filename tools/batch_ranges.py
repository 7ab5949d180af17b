#!/usr/bin/env python3
"""Per-launch kernel times of a multi-start batch (P searches of the same problem, different streams) by generation range -- HIP events around every launch
(the loop is slower than untimed; read the split).

    python tools/batch_ranges.py [P] [lazy 0|1] [ants] [grid]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import api, synth  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lazy = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
    ants = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    s = api.AcsSolver(ctx, grid, n_slots=P, max_colony=ants, lazy=lazy)
    p = api.default_params(max_iteration=500, predict=3.0 * n * ants / 256, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=1)
    for rep in range(2):
        s.init_pheromone(1.0)
        s.begin(p, [ids[0]] * P, [ids[1]] * P, streams=list(range(100, 100 + P)))
        ctx.sync()
        for label, gens in (("0-4", 5), ("5-19", 15), ("20-49", 30), ("50-99", 50), ("100-499", 400)):
            s.profile(True, 1)
            s.run(gens)
            s.sync()
            r = s.profile_read()
            if rep == 1:
                print("%s P %d generations %-8s groups %d  " % ("lazy " if lazy else "dense", P, label, s.pipeline_groups()) +
                      "  ".join("%s %.1f us x %d" % (k, v["ms"] / max(1, v["launches"]) * 1e3, v["launches"]) for k, v in r.items() if v["launches"]), flush=True)


if __name__ == "__main__":
    main()
