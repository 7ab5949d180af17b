#!/usr/bin/env python3
"""Walk-kernel time per generation against the LONGEST walk of that generation (diagnostic).

A walk launch lasts as long as its slowest wavefront, so the figure that matters for the exploratory
generations is  launch time / max steps of any ant  = time of one general (non-replay) step of a lone wave.

    python tools/walk_steps.py [generations] [grid] [ants]      (WA_REPLAY=0 to see the general loop only)
"""
import os
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from welding_robot_amd import api, synth


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ants = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    s = api.AcsSolver(ctx, grid, 1, ants)
    p = api.default_params(max_iteration=G, predict=float(ants / 0.35), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=12345)
    for rep in range(2):   # first pass warms the allocator / code objects
        s.init_pheromone(1.0)
        s.begin(p, ids[0], ids[1])
        rows = []
        for g in range(G):
            s.profile(True, 1)
            s.run(1)
            pr = s.profile_read()
            L, lens = s.ants()
            rows.append((g, pr["walk"]["ms"] * 1e3, int(lens.max()) - 1, float(lens.mean()) - 1, pr["evaporate"]["ms"] * 1e3, pr["deposit"]["ms"] * 1e3))
    tot = 0.0
    for g, us, mx, mean, ev, dp in rows:
        tot += us
        if g < 12 or g % 10 == 0:
            print("gen %3d: walk %7.1f us  max steps %5d  mean %6.1f  -> %6.1f ns per step of the longest walk; sweep+rank+mark %5.1f us, apply+table %5.1f us" % (
                g, us, mx, mean, 1e3 * us / max(mx, 1), ev, dp))
    print("total walk over %d generations: %.2f ms; best %g" % (G, tot / 1e3, float(s.trace()["bestL"][-1])))


if __name__ == "__main__":
    main()
