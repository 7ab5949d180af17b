#!/usr/bin/env python3
"""One or several 256-ant searches of an n^3 grid (6 or 26 neighbours), generations 5..79: whole-generation time and per-launch kernel times
under the sweep cache policy WA_SWEEP_NT selects (unset: the library's rule) -- the measurements behind sweep_policy() (csrc/host_acs.inc),
profiles/r04/sweep_policy_in_loop.txt.      python tools/sweep_policy_in_loop.py <n> <searches> [6|26]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from welding_robot_amd import api, synth
n = int(sys.argv[1]); P = int(sys.argv[2]); nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
s = api.AcsSolver(ctx, grid, n_slots=P, max_colony=256, neighbourhood=nb)
gens = 80
p = api.default_params(max_iteration=gens, predict=731.43 * n / 128, fixed_colony=256, rng_mode=api.RNG_DEV, seed=4242)
s.init_pheromone(1.0)
s.begin(p, [ids[0]] * P, [ids[1]] * P, streams=list(range(P)))
s.run(5); s.sync()
s.profile(True, 5)
t0 = time.perf_counter(); s.run(gens - 5); s.sync(); dt = time.perf_counter() - t0
pr = s.profile_read()
print("NT=%s n=%d P=%d nb=%d: %.1f us per generation; walk %.1f fused %.1f apply %.1f" % (os.environ.get("WA_SWEEP_NT", "rule"), n, P, nb, dt * 1e6 / (gens - 5),
      *[pr[k]["ms"] / max(pr[k]["launches"], 1) * 1e3 for k in ("walk", "evaporate", "deposit")]))
