#!/usr/bin/env python3
"""Per-generation walk time of the 26-neighbour variant (diagnostic).   python tools/walk_profile26.py [generations]"""
import os, sys
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
s = api.AcsSolver(ctx, grid, 1, 256, neighbourhood=26)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 300
p = api.default_params(max_iteration=G, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
s.begin(p, 16513, 2097151)
w, mx = [], []
for g in range(G):
    s.profile(True, 1)
    s.run(1)
    pr = s.profile_read()
    _, lens = s.ants()
    w.append(pr["walk"]["ms"] * 1e3)
    mx.append(int(lens.max()) - 1)
w, mx = np.array(w), np.array(mx)
t = s.trace()
for lo, hi in [(0, 10), (10, 20), (20, 40), (40, 60), (60, 80), (80, 100), (100, 150), (150, G)]:
    if lo < G:
        hi = min(hi, G)
        print("gens %3d-%3d: walk %7.1f us avg (max %7.1f), longest walk %6.1f steps avg -> %6.1f ns per step of it, steps/ant %6.1f, bestL %.3f" % (
            lo, hi, w[lo:hi].mean(), w[lo:hi].max(), mx[lo:hi].mean(), 1e3 * w[lo:hi].sum() / mx[lo:hi].sum(), t["steps"][lo:hi].mean() / 256, t["bestL"][hi - 1]))
print("total walk %.1f ms over %d generations" % (w.sum() / 1e3, G))
