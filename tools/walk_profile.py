#!/usr/bin/env python3
"""Walk time by generation range over a run (diagnostic): where do the 500 generations spend their walk time?
    python tools/walk_profile.py [generations] [--stepwise]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
s = api.AcsSolver(ctx, grid, 1, 256)
G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 500
p = api.default_params(max_iteration=G, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
s.begin(p, 16513, 2097151)
RANGES = [(0, 10), (10, 20), (20, 40), (40, 60), (60, 80), (80, 100), (100, 150), (150, 250), (250, G)]
if "--stepwise" in sys.argv:   # one wa_acs_run call per generation: per-generation maxima, but the straggler hand-over never engages
    w = []
    for g in range(G):
        s.profile(True, 1)
        s.run(1)
        pr = s.profile_read()
        w.append(pr["walk"]["ms"] * 1e3)
    w = np.array(w)
    t = s.trace()
    for lo, hi in RANGES:
        if lo < G:
            hi = min(hi, G)
            print("gens %3d-%3d: walk %7.1f us avg (max %7.1f), steps/ant %6.1f, bestL %.0f, share of total walk %.1f%%" % (
                lo, hi, w[lo:hi].mean(), w[lo:hi].max(), t["steps"][lo:hi].mean() / 256, t["bestL"][hi - 1], 100 * w[lo:hi].sum() / w.sum()))
    print("total walk %.1f ms over %d generations (one call per generation: no hand-over)" % (w.sum() / 1e3, G))
else:                          # one call per range, an event pair around every launch: the product's behaviour (only a call's last generation hands nothing over)
    rows = []
    for lo, hi in RANGES:
        if lo < G:
            hi = min(hi, G)
            s.profile(True, 1)
            s.run(hi - lo)
            pr = s.profile_read()
            rows.append((lo, hi, pr["walk"]["ms"] * 1e3))
    t = s.trace()
    tot = sum(r[2] for r in rows)
    for lo, hi, us in rows:
        print("gens %3d-%3d: walk %7.1f us avg, steps/ant %6.1f, bestL %.0f, share of total walk %.1f%%" % (
            lo, hi, us / (hi - lo), t["steps"][lo:hi].mean() / 256, t["bestL"][hi - 1], 100 * us / tot))
    print("total walk %.1f ms over %d generations (WA_STRAGGLERS=%s)" % (tot / 1e3, G, os.environ.get("WA_STRAGGLERS", "1")))
