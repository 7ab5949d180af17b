#!/usr/bin/env python3
"""Random searches solved with the straggler hand-over on and off: which ants are handed over depends on timing, nothing
observable may -- trace incl. steps and finite ants, the last generation's ants, the whole field.  120 trials per call:
    [NB=26] [T0=first trial number] python tools/straggler_stress.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import api, synth
ctx = api.Context(0)
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0; total_h = 0
for trial in range(int(os.environ.get("T0", "100")), int(os.environ.get("T0", "100")) + 120):
    rs = np.random.RandomState(trial)
    n = int(rs.choice([24, 32, 48, 64, 96]))
    occ = float(rs.choice([0.0, 0.05, 0.1, 0.2, 0.3]))
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=100 + trial, occ_prob=occ)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    ants = int(rs.choice([8, 16, 37, 64, 128, 200, 256])); gens = int(rs.randint(2, 50))
    fixed = ants if rs.rand() < 0.7 else 0
    predict = 3.0 * n if fixed else float(ants / 0.35 * prec)
    res = {}
    for mode in ("1", "0"):
        os.environ["WA_STRAGGLERS"] = mode
        s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants, neighbourhood=int(os.environ.get("NB", "6")))
        p = api.default_params(max_iteration=gens, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_DEV, seed=1000 + trial,
                               rho=[0.8, 0.5, 0.95][trial % 3], beta=[0.6, 1.0, 2.0][(trial // 3) % 3])
        out = np.zeros(16, np.uint64)
        s.init_pheromone(1.0)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        s.solve(p, ids[0], ids[1], streams=[trial & 7])
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
        t = s.trace(); L, lens = s.ants()
        res[mode] = (t["steps"].copy(), t["finite"].copy(), bits(t["bestL"]).copy(), t["colony"].copy(), bits(L).copy(), lens.copy(), bits(s.pheromone()).copy(), int(out[9]), int(out[7]))
        s.close()
    grid.close()
    same = all(np.array_equal(a, b) for a, b in zip(res["1"][:7], res["0"][:7]))
    h, r = res["1"][7], res["1"][8]
    total_h += h
    if not same or h != r or res["0"][7] != 0:
        bad += 1
        print("MISMATCH trial", trial, n, occ, ants, gens, "handed", h, "resumed", r)
print("trials 120, mismatches", bad, "ants handed over in total", total_h)
