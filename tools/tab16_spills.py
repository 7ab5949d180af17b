"""Round 6: 16-bit tabu entries on the 256^3 pair-planning batch (120 pairs x 24 ants x 150 generations): walk geometry, time, walks spilled to the
bitmap and how many of them because a 16-bit entry found no slot within 14 (wa_acs_debug_counters [13], [14]; knobs build).  profiles/r06/tab16.txt"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'examples')
import numpy as np
from welding_robot_amd import api, synth
import plan_batch
ctx = api.Context(0)
n = 256
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, 16, seed=7)
pairs = [(i, j) for i in range(16) for j in range(i + 1, 16)]
for mode in ("-1", "0"):
    os.environ["WA_TAB16"] = mode
    s = api.AcsSolver(ctx, grid, n_slots=120, max_colony=24, lazy=True)
    p = api.default_params(max_iteration=150, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
    out = np.zeros(16, np.uint64)
    s.init_pheromone(1.0)
    for rep in range(2):
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        t0 = time.perf_counter()
        s.solve(p, [int(pts[i]) for i, j in pairs], [int(pts[j]) for i, j in pairs], streams=list(range(len(pairs))))
        ctx.sync()
        dt = time.perf_counter() - t0
        s.reset_pheromone(1.0)
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
    steps = sum(int(s.trace(q)["steps"].sum()) for q in range(len(pairs)))
    print("WA_TAB16=%s: %s, %.4f s, walks spilled %d, of them by 16-bit overflow %d, ant walks %d" % (mode, s.walk_info(), dt, int(out[13]), int(out[14]), 120 * 24 * 150), flush=True)
    s.close()
