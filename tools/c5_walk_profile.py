#!/usr/bin/env python3
"""BASELINE config C5, one batch of pair searches (96 slots x 24 ants on 256^3, lazy evaporation): kernel time per
generation range (diagnostic) -- where do the 150 generations of a batch spend their time?

    python tools/c5_walk_profile.py [grid] [points] [slots] [generations] [batch]
"""
import os, sys
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 96
G = int(sys.argv[4]) if len(sys.argv) > 4 else 150
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, P, seed=7)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 10          # which batch of the end-point-ordered list (as plan_batch.py runs them)
allp = sorted([(i, j) for i in range(P) for j in range(i + 1, P)], key=lambda t: (t[1], t[0]))
pairs = allp[B * slots:(B + 1) * slots]
s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
p = api.default_params(max_iteration=G, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
s.begin(p, [pts[a] for a, b in pairs], [pts[b] for a, b in pairs], streams=list(range(len(pairs))))
rows = []
for g in range(G):
    s.profile(True, 1)
    s.run(1)
    pr = s.profile_read()
    rows.append((pr["walk"]["ms"] * 1e3, pr["evaporate"]["ms"] * 1e3, pr["deposit"]["ms"] * 1e3))
w = np.array(rows)
for lo, hi in [(0, 10), (10, 20), (20, 40), (40, 60), (60, 80), (80, 100), (100, 150)]:
    if lo < G:
        hi = min(hi, G)
        print("gens %3d-%3d: walk %7.1f us avg (max %7.1f)  sweep+rank+mark %6.1f  apply+table %6.1f   share of batch %.1f%%" % (
            lo, hi, w[lo:hi, 0].mean(), w[lo:hi, 0].max(), w[lo:hi, 1].mean(), w[lo:hi, 2].mean(), 100 * w[lo:hi].sum() / w.sum()))
print("batch of %d searches: walk %.1f ms, sweep+rank+mark %.1f ms, apply+table %.1f ms over %d generations" % (
    len(pairs), w[:, 0].sum() / 1e3, w[:, 1].sum() / 1e3, w[:, 2].sum() / 1e3, G))
