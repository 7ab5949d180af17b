#!/usr/bin/env python3
"""Does a saturated batch of pair searches give the same answer every time?  The same SLOTS lazily evaporating 24-ant searches on an N^3 grid are solved RUNS
times on one solver (reset in between); every search's per-generation trace must equal the first run's.  Used with variant builds that change the
scheduling of the post-walk launch (-DWA_RANK_LDS=64: eight instead of four of its blocks per CU; WA_LAZY_BLOCKS=2: two long background blocks per
search) this is what exposed the publishing block's late read of the global best in round 6 (profiles/r06/best_copy_race.txt); tools/state_hash.py then
names the launch and the array.

    [WELDACS_LIB=build/libweldacs_rank64.so] [WA_LAZY_BLOCKS=2] N=128 SLOTS=224 GENS=60 RUNS=40 G=1 [COLONY=24 DENSE=0 NB=6 POINTS=64] python tools/determinism_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from welding_robot_amd import api, synth
n = int(os.environ.get("N", "128")); slots = int(os.environ.get("SLOTS", "224")); gens = int(os.environ.get("GENS", "60")); runs = int(os.environ.get("RUNS", "20"))
G = int(os.environ.get("G", "1"))
colony = int(os.environ.get("COLONY", "24")); dense = os.environ.get("DENSE", "0") == "1"; nb = int(os.environ.get("NB", "6")); npts = int(os.environ.get("POINTS", "64"))
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, npts, seed=7)
pairs = [(i, j) for i in range(npts) for j in range(i + 1, npts)][:slots]
slots = len(pairs)
a, b = [int(pts[i]) for i, _ in pairs], [int(pts[j]) for _, j in pairs]
p = api.default_params(max_iteration=gens, predict=colony / 0.35, rng_mode=api.RNG_DEV, seed=7)
ref = None
s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=colony, lazy=not dense, neighbourhood=nb)
ctx.check(ctx.lib.wa_acs_set_pipeline(s.h, G))
nbad = 0
for rep in range(runs):
    s.solve(p, a, b, streams=list(range(slots)))
    tr = [s.trace(q) for q in range(slots)]
    cur = [(t["steps"].copy(), t["bestL"].copy(), t["iterbestL"].copy(), t["finite"].copy() if "finite" in t else None) for t in tr]
    if ref is None:
        ref = cur
    else:
        for q in range(slots):
            if not np.array_equal(ref[q][0], cur[q][0]):
                g = int(np.argmax(ref[q][0] != cur[q][0]))
                nbad += 1
                print("run", rep, "slot", q, "first differing generation", g, "steps", int(ref[q][0][g]), int(cur[q][0][g]), "iterbest", float(ref[q][2][g]), float(cur[q][2][g]),
                      "best before", float(ref[q][1][g - 1]) if g else None, float(cur[q][1][g - 1]) if g else None, "pair", pairs[q], flush=True)
    s.reset_pheromone(1.0)
print("lib", os.environ.get("WELDACS_LIB", "product"), "dense" if dense else "lazy", "nb", nb, "colony", colony, "grid", n, "groups", G, "runs", runs, "slot-runs that differ:", nbad, "of", (runs - 1) * slots, flush=True)
