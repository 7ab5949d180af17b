#!/usr/bin/env python3
"""hipGraph replay of the generation loop (WA_GRAPH=G): generations/s with the per-dispatch profiling OFF, at the
benchmark size and at the reference's small demo sizes, where the loop is launch-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api
import oracle_lib as O

ctx = api.Context(0)
G = os.path.join(ROOT, "tests", "golden")
cases = []
og = O.synth_grid(128, seed=2024, occ_prob=0.10)
cases.append(("C3 128^3 256 ants", og, 16513, 2097151, 500, 731.43, 256))
og2 = O.grid_from_mesh(O.stl_parse(open(os.path.join(G, "simplified_piece.stl"), "rb").read()), 0.0148, 4)
cases.append(("C2 64x33x23 128 ants", og2, 2177, 48575, 200, 5.4126, 128))
og1 = O.grid_from_mesh(O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read()), 0.0219, 8)
cases.append(("C1 25x32x25 16 ants", og1, 4130, 17521, 150, 1.03, 16))
for name, g, sid, eid, gens, predict, ants in cases:
    dg = api.Grid.from_occupancy(ctx, g.free, g.cx, g.cy, g.cz, g.precision, g.wall)
    ref = None
    for glen in (0, 4, 10, 50):
        os.environ["WA_GRAPH"] = str(glen)
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ants)
        p = api.default_params(max_iteration=gens, predict=predict, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=12345)
        s.solve(p, sid, eid); s.reset_pheromone(1.0)
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.perf_counter(); s.solve(p, sid, eid); t = time.perf_counter() - t0
            best = min(best, t)
            if rep < 2:
                s.reset_pheromone(1.0)
        tr = s.trace(); ph = s.pheromone()
        key = (tr["bestL"].tobytes(), tr["steps"].tobytes(), ph.tobytes())
        if ref is None:
            ref = key
        print("%-22s WA_GRAPH=%-2d %4d generations: %7.2f ms = %8.0f gen/s (%.1f us/generation)  equal to plain launches: %s" % (
            name, glen, gens, best * 1e3, gens / best, best / gens * 1e6, key == ref))
        s.close()
    dg.close()
