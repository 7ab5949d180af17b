#!/usr/bin/env python3
"""Would handing stragglers over pay for LAZY pair-planning batches (BASELINE config C5: 256^3, 24 ants, 150 generations)?  VERDICT r05 task 8.
An ant that is already longer than floor(lambda - 1) + 1 arrivals of its own generation can neither deposit (ACSRank_3D.hpp:200) nor
become the best path (:263-264).  One batch of C5's pair searches run generation by generation on the device, every ant's node count
read back after every generation; per generation range:
  critical : how much of a batch launch's CRITICAL PATH sits behind the cut -- 1 - max over searches of (the cut_n-th shortest arrival,
             or the longest ant if fewer arrive) / max over all ants of the batch (a walk launch lasts as long as its longest walk);
  steps    : how many of the batch's ant STEPS lie behind the cut (what a hand-over would move beside the next generation's walks -- it
             removes no work: every walk stays complete).
    python tools/straggler_lazy.py [slots] [generations]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from welding_robot_amd import api, synth  # noqa: E402


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 224
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    n, P = 256, 64
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(free, n, P, seed=7)
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    rs = np.random.RandomState(3)
    idx = sorted(rs.choice(len(pairs), slots, replace=False).tolist(), key=lambda k: pairs[k][1])   # a batch as the plan forms it: mixed lengths, end-point order
    s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
    p = api.default_params(max_iteration=gens, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
    s.init_pheromone(1.0)
    s.begin(p, [int(pts[pairs[k][0]]) for k in idx], [int(pts[pairs[k][1]]) for k in idx], streams=idx)
    rows = []
    for g in range(gens):
        s.run(1)
        crit_all, crit_cut, steps_all, steps_behind = 0, 0, 0, 0
        for q in range(slots):
            L, lens = s.ants(q)
            colony = lens.size
            cut_n = max(1, int(0.2 * colony - 1.0) + 1)
            arrived = np.sort(lens[np.isfinite(L)])
            longest = int(lens.max())
            lim = int(arrived[cut_n - 1]) if arrived.size >= cut_n else longest      # node count beyond which an ant cannot matter any more
            crit_all = max(crit_all, longest)
            crit_cut = max(crit_cut, lim)
            steps_all += int(lens.sum())
            steps_behind += int(np.maximum(lens - lim, 0).sum())
        rows.append((g, crit_all, crit_cut, steps_all, steps_behind))
    print("# tools/straggler_lazy.py: %d concurrent lazy pair searches (256^3, 24 ants), %d generations, cut = floor(lambda - 1) + 1 arrivals" % (slots, gens))
    for lo, hi in ((0, 10), (10, 20), (20, 40), (40, 80), (80, 150)):
        sel = [r for r in rows if lo <= r[0] < hi]
        if not sel:
            continue
        ca, cc = sum(r[1] for r in sel), sum(r[2] for r in sel)
        sa, sb = sum(r[3] for r in sel), sum(r[4] for r in sel)
        print("generations %3d-%3d: longest walk %6.0f nodes per generation, behind the cut %5.1f %% of the launches' critical path, %5.1f %% of the ant steps (%.0f steps per generation)"
              % (lo, hi - 1, ca / len(sel), 100.0 * (1 - cc / ca), 100.0 * sb / sa, sa / len(sel)))
    ca, cc = sum(r[1] for r in rows), sum(r[2] for r in rows)
    sa, sb = sum(r[3] for r in rows), sum(r[4] for r in rows)
    print("all generations: %5.1f %% of the critical path, %5.1f %% of the steps" % (100.0 * (1 - cc / ca), 100.0 * sb / sa))
    print(json.dumps(dict(slots=slots, generations=gens, critical_path_behind_cut=1 - cc / ca, steps_behind_cut=sb / sa)))


if __name__ == "__main__":
    main()
