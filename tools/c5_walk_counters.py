#!/usr/bin/env python3
"""What bounds the walk of a C5 batch (256^3, 224 concurrent 24-ant pair searches, lazy evaporation, 150 generations)?  The workload for the counter
passes of tools/c5_walk_counters.sh: ONE batch (after an untimed one that brings the solver into its steady state), its wall time and its step count.

    python tools/c5_walk_counters.py [slots] [grid]            (summary mode: python tools/c5_walk_counters.py --summary <rocprof dirs...>)"""
import collections
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summary(dirs):
    """totals per kernel: duration from the kernel traces, every counter found in the counter files"""
    for d in dirs:
        tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
        dur, nd = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                nd[k] += 1
        print("## %s" % os.path.basename(d.rstrip("/")))
        for k in sorted(dur, key=lambda k: -dur[k])[:6]:
            print("%-60s dispatches %5d  total %10.3f ms" % (k[:60], nd[k], dur[k] / 1e6))
        for (k, c) in sorted(tot, key=lambda kc: (kc[1], -tot[kc])):
            if tot[(k, c)] > 0 and cnt[(k, c)] >= 10:
                print("%-28s %-60s dispatches %5d  total %.6g" % (c, k[:60], cnt[(k, c)], tot[(k, c)]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summary":
        return summary(sys.argv[2:])
    import numpy as np
    from welding_robot_amd import api, synth
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 224
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    gens = 150
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(free, n, 64, seed=7)
    pairs = [(i, j) for i in range(64) for j in range(i + 1, 64)][:slots]
    order = os.environ.get("C5W_ORDER", "")          # experiment: the searches of the batch by descending / ascending distance of their end points
    if order:
        xyz = np.stack([np.asarray(pts) % n, (np.asarray(pts) // n) % n, np.asarray(pts) // (n * n)], 1).astype(np.int64)
        dist = lambda ij: int(np.abs(xyz[ij[0]] - xyz[ij[1]]).sum())
        pairs.sort(key=dist, reverse=(order == "desc"))
        if order == "interleave":      # long searches first WITHIN each of the two pipelined groups (group g = slots [g * slots/2, (g+1) * slots/2))
            pairs.sort(key=dist, reverse=True)
            pairs = pairs[0::2] + pairs[1::2]
    s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
    p = api.default_params(max_iteration=gens, predict=24 / 0.35, rng_mode=api.RNG_DEV, seed=7)
    a, b = [int(pts[i]) for i, _ in pairs], [int(pts[j]) for _, j in pairs]
    s.solve(p, a, b, streams=list(range(slots)))
    s.reset_pheromone(1.0)
    ctx.sync()
    t0 = time.perf_counter()
    s.begin(p, a, b, streams=list(range(slots)))
    s.run(gens)
    t_enq = time.perf_counter() - t0
    s.sync()
    t = time.perf_counter() - t0
    steps = sum(int(s.trace(q)["steps"].sum()) for q in range(slots))
    print("order %r: %d^3, %d slots x 24 ants, %d generations (the SECOND of two identical batches): %.2f ms (the host had enqueued it after %.2f ms), %d ant steps = %.3f G steps/s; walk geometry %s"
          % (order, n, slots, gens, t * 1e3, t_enq * 1e3, steps, steps / t / 1e9, s.walk_info()), flush=True)
    print("   (both batches run the same %d steps: counter totals of the whole process / 2 = one batch)" % steps)


if __name__ == "__main__":
    main()
