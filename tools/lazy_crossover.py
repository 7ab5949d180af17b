#!/usr/bin/env python3
"""Where does lazy evaporation (wa_acs_create_lazy) start to pay for PAIR PLANNING (24 ants, 150 generations per pair search -- the shape of
ACS_Rank::searchBestPathOfPoints, ACSRank_3D.hpp:427-504)?  VERDICT r05 weak item 6: a lone lazy search is slower than a lone dense one
(walk 117 vs 76 us at 128^3 / 256 ants) while the drop-in used lazy for every DEV-mode pair job.  For grids of 32^3 .. 256^3 and 1 .. 21
pairs: seconds per plan (solver creation excluded, best of 3), dense and lazy, same costs.  -> profiles/r06/lazy_crossover.txt

    python tools/lazy_crossover.py [--grids 32,64,128,192,256] [--points 2,3,4,5,6,7]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np  # noqa: E402

from welding_robot_amd import api, synth  # noqa: E402
import plan_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="32,64,128,192,256")
    ap.add_argument("--points", default="2,3,4,5,6,7")
    ap.add_argument("--gens", type=int, default=150)
    a = ap.parse_args()
    ctx = api.Context(0)
    predict = float(24 / 0.35)
    for n in [int(x) for x in a.grids.split(",")]:
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        for K in [int(x) for x in a.points.split(",")]:
            pts = synth.synth_weld_points(free, n, K, seed=7)
            pairs = K * (K - 1) // 2
            res = {}
            ref = None
            for lazy in (False, True):
                best = 1e9
                for rep in range(3):
                    t0 = time.perf_counter()
                    cost, paths, _ = plan_batch.plan(ctx, grid, pts, a.gens, predict, 7, pairs, lazy=lazy)
                    ctx.sync()
                    best = min(best, time.perf_counter() - t0 - plan_batch.plan.last_create_s)
                if ref is None:
                    ref = cost
                res["lazy" if lazy else "dense"] = best
                same = bool(np.array_equal(cost, ref))
            print(json.dumps(dict(grid=n, voxels=n ** 3, pairs=pairs, slots_x_voxels=pairs * n ** 3, dense_ms=round(res["dense"] * 1e3, 2), lazy_ms=round(res["lazy"] * 1e3, 2),
                                  lazy_over_dense=round(res["lazy"] / res["dense"], 3), same_costs=same)), flush=True)
        grid.close()
    ctx.close()


if __name__ == "__main__":
    main()
