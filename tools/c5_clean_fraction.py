#!/usr/bin/env python3
"""Diagnostic for the C5 walk: what fraction of an ant's steps happens in 8^3 bricks that hold no deposited voxel (nor one within
a voxel of the brick)?  One pair search of the C5 shape (256^3, 24 ants), generation by generation; a voxel is deposited once
one of the generation's ranked ants (the 5 shortest finite paths) walked over it.

    python tools/c5_clean_fraction.py [grid] [generations]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from welding_robot_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = 8
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
pts = synth.synth_weld_points(free, n, 64, seed=7)
s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=24, lazy=True)
p = api.default_params(max_iteration=G, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
s.begin(p, [pts[3]], [pts[40]], streams=[100])
nb = (n + B - 1) // B
dirty = np.zeros((nb, nb, nb), bool)          # bricks with a deposited voxel in them or within one voxel
def bricks_with_halo(ids):
    z, r = np.divmod(ids, n * n); y, x = np.divmod(r, n)
    out = set()
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                zz, yy, xx = np.clip(z + dz, 0, n - 1) // B, np.clip(y + dy, 0, n - 1) // B, np.clip(x + dx, 0, n - 1) // B
                out.update(zip(zz.tolist(), yy.tolist(), xx.tolist()))
    return out
for g in range(G):
    s.run(1)
    L, lens = s.ants()
    paths = [s.ant_path(a) for a in range(len(lens))]
    tot = clean = 0
    for ids in paths:
        z, r = np.divmod(ids, n * n); y, x = np.divmod(r, n)
        c = ~dirty[z // B, y // B, x // B]
        tot += len(ids); clean += int(c.sum())
    order = np.argsort(L, kind="stable")
    for a in order[:5]:
        if np.isfinite(L[a]):
            for t in bricks_with_halo(paths[a]):
                dirty[t] = True
    if g < 10 or g % 5 == 0:
        print("generation %3d: %6d steps, %5.1f %% of them in clean bricks; dirty bricks now %5.1f %%" % (g, tot, 100.0 * clean / max(tot, 1), 100.0 * dirty.mean()))
