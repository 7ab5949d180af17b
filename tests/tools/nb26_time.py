#!/usr/bin/env python3
"""26-neighbour variant at the benchmark size (128^3 synthetic grid, 256 ants): generations/s and per-kernel time,
beside the 6-neighbour path on the same grid, and the CPU port for a few generations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api, synth
import oracle_lib as O

N, ANTS, GENS = 128, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = api.Context(0)
og = O.synth_grid(N, seed=2024, occ_prob=0.10)
dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, 0)
sid, eid = 16513, 2097151
for nb in (6, 26):
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ANTS, neighbourhood=nb)
    p = api.default_params(max_iteration=GENS, predict=731.43, fixed_colony=ANTS, rng_mode=api.RNG_DEV, seed=12345)
    s.solve(p, sid, eid); s.reset_pheromone(1.0)        # warm-up
    s.profile(True, 1)
    t0 = time.perf_counter(); s.solve(p, sid, eid); t1 = time.perf_counter()
    pr = s.profile_read()
    cost, path, ch = s.result()
    tr = s.trace()
    per = ", ".join("%s %.1f us" % (k, v["ms"] / v["launches"] * 1e3) for k, v in pr.items() if v["launches"])
    sweep_s = pr["evaporate"]["ms"] / max(pr["evaporate"]["launches"], 1) * 1e-3
    sweep_bytes = 8 * nb * N ** 3
    print("nb=%2d: %d generations in %.1f ms = %.0f gen/s (per-dispatch events on: %s); sweep %.0f MB -> %.2f TB/s; best cost %.4f in %d nodes, "
          "first generation with that cost %d, steps/ant gen0 %.0f, mean %.0f" % (
              nb, GENS, (t1 - t0) * 1e3, GENS / (t1 - t0), per, sweep_bytes / 1e6,
              sweep_bytes / sweep_s / 1e12, cost, len(path),
              int(np.argmax(tr["bestL"] == tr["bestL"][-1])), tr["steps"][0] / ANTS, tr["steps"].mean() / ANTS))
    if nb == 26:
        a = O.Acs(og, nb=26)
        g = 4
        t0 = time.perf_counter(); otr = a.solve(sid, eid, g, 731.43, fixed_colony=ANTS, mode=O.DEV, seed=12345); t1 = time.perf_counter()
        print("        CPU port (1 thread) first %d generations: %.2f gen/s; cost trace equal: %s" % (
            g, g / (t1 - t0), bool(np.array_equal(otr["bestL"].view(np.uint32), tr["bestL"][:g].view(np.uint32)))))
    s.close()
