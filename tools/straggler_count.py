#!/usr/bin/env python3
"""How many ants of BASELINE config C3 are handed over as stragglers (wa_acs_debug_counters: [9] handed over, [7] finished by a resume
block) and what the walk launch costs with and without the mechanism:   python tools/straggler_count.py   /   WA_STRAGGLERS=0 python tools/straggler_count.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from welding_robot_amd import api, synth
from welding_robot_amd import dist as wd
n, ants = 128, 256
ctx = api.Context(0)
wl = wd.per_rank_workload(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=wl["grid_seed"], occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants)
p = api.default_params(max_iteration=40, predict=3.0 * n, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=wl["rng_seed"])
out = np.zeros(16, np.uint64)
for rep in range(2):
    s.init_pheromone(1.0)
    s.begin(p, ids[0], ids[1], streams=[wl["stream"]])
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
    s.profile(True, 1)
    s.run(40); s.sync()
ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
pr = s.profile_read()
print("WA_STRAGGLERS=%s: walk %.1f us per generation over 40 generations, handed over %d, resumed %d" % (os.environ.get("WA_STRAGGLERS", "1"), pr["walk"]["ms"] / 40 * 1e3, int(out[9]), int(out[7])))
