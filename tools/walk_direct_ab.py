#!/usr/bin/env python3
"""A/B of the walk loop's variants on the saturated workloads (round 5): look-ahead without touch loads (WA_WALK_DIRECT=0) against the
loop without look-ahead (WA_WALK_DIRECT=1), optionally across tabu-table sizes (WA_HASH_LOG2).  One process, solvers created one after
another (the knobs are read by wa_acs_create; the context's block cache makes the re-creation cheap).

  python tools/walk_direct_ab.py [--c5] [--ms 8,16] [--kinds dense,lazy] [--hash 0,11,12,13] [--reps 2] [--gens 100]

  --c5     BASELINE config C5: 2 016 pair searches x 150 generations on 256^3 (lazy evaporation), seconds per plan
  --ms     multi-start batches: P independent 128^3 / 256-ant searches, problem-generations/s (generations 5..gens-1)
"""
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


def setenv(knobs):
    for k in ("WA_WALK_DIRECT", "WA_HASH_LOG2", "WA_WALK_WARM", "WA_WALK_LDS_PAD"):
        os.environ.pop(k, None)
    for k, v in knobs.items():
        if v is not None:
            os.environ[k] = str(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5", action="store_true")
    ap.add_argument("--ms", default="")
    ap.add_argument("--kinds", default="dense,lazy")
    ap.add_argument("--groups", default="1,2")
    ap.add_argument("--hash", default="0")
    ap.add_argument("--direct", default="0,1")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pad", default="0", help="WA_WALK_LDS_PAD values (bytes of unused dynamic LDS per walk block: occupancy at a constant table)")
    ap.add_argument("--gens", type=int, default=100)
    ap.add_argument("--batches", action="store_true", help="--c5: seconds per batch (solve, reset, read-back)")
    ap.add_argument("--points", type=int, default=64)
    ap.add_argument("--grid", type=int, default=256)
    a = ap.parse_args()
    ctx = api.Context(0)
    hashes = [int(x) for x in a.hash.split(",")]
    directs = [int(x) for x in a.direct.split(",")]
    pads = [int(x) for x in a.pad.split(",")]
    if a.c5:
        import plan_batch
        n = a.grid
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        pts = synth.synth_weld_points(free, n, a.points, seed=7)
        predict = float(0.35 ** -1 * 24)
        ref = None
        for h in hashes:
            for d, pad in [(d, pad) for d in directs for pad in pads]:
                setenv(dict(WA_WALK_DIRECT=d, WA_HASH_LOG2=h or None, WA_WALK_LDS_PAD=pad or None))
                ts = []
                for r in range(a.reps):
                    t0 = time.perf_counter()
                    cost, paths, _ = plan_batch.plan(ctx, grid, pts, 150, predict, 7, 0, lazy=True)
                    ts.append(time.perf_counter() - t0 - plan_batch.plan.last_create_s)
                    if a.batches:
                        print(json.dumps(dict(what="c5 batches", direct=d, hash_log2=h or "rule", rep=r, solve_reset_read_s=plan_batch.plan.last_batch_s)), flush=True)
                if ref is None:
                    ref = cost
                print(json.dumps(dict(what="c5", grid=n, points=a.points, direct=d, hash_log2=h or "rule", lds_pad=pad, slots=plan_batch.plan.last_slots, t_pairs_s=[round(t, 4) for t in ts],
                                      create_s=round(plan_batch.plan.last_create_s, 3), same_costs=bool(np.array_equal(cost, ref)))), flush=True)
        grid.close()
    if a.ms:
        import bench
        n = 128
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
        for kind in a.kinds.split(","):
            for P in [int(x) for x in a.ms.split(",")]:
                ref = None
                for G in [int(x) for x in a.groups.split(",")]:
                    for h in hashes:
                        for d, pad in [(d, pad) for d in directs for pad in pads]:
                            setenv(dict(WA_WALK_DIRECT=d, WA_HASH_LOG2=h or None, WA_WALK_LDS_PAD=pad or None))
                            vals = []
                            for r in range(a.reps):
                                out, hist, steps, _ = bench.multi_start_run(ctx, grid, ids, n, 256, P, G, a.gens, kind == "lazy", 5)
                                vals.append(out["problem_generations_per_s"])
                                if ref is None:
                                    ref = (hist, steps)
                            k = out["kernel_ms_per_launch"]
                            print(json.dumps(dict(what="multi_start", kind=kind, P=P, G=G, direct=d, hash_log2=h or "rule", lds_pad=pad, pgps=[round(v) for v in vals],
                                                  walk_us=round(1e3 * k["walk"], 1), sweep_us=round(1e3 * k["evaporate"], 1),
                                                  same=bool(np.array_equal(hist, ref[0]) and np.array_equal(steps, ref[1])))), flush=True)
        grid.close()
    ctx.close()


if __name__ == "__main__":
    main()
