#!/usr/bin/env python3
"""A/B timing of walk-loop variants (diagnostic): every library given on the command line runs the first G generations of
BASELINE config C3 in its own process (WELDACS_LIB), several rounds interleaved, and must reproduce the first library's
per-generation trace bit for bit.  Prints the walk time summed over the generations and per step of the longest walk.

    python tools/walk_ab.py [--gens 12] [--rounds 3] lib_a.so lib_b.so ...
"""
import json
import os
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(G):
    import numpy as np
    from welding_robot_amd import api, synth
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    s = api.AcsSolver(ctx, grid, 1, 256)
    p = api.default_params(max_iteration=G, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    best = None
    for rep in range(3):
        s.init_pheromone(1.0)
        s.begin(p, 16513, 2097151)
        us, steps = [], []
        for g in range(G):
            s.profile(True, 1)
            s.run(1)
            pr = s.profile_read()
            _, lens = s.ants()
            us.append(pr["walk"]["ms"] * 1e3)
            steps.append(int(lens.max()) - 1)
        if best is None or sum(us) < sum(best[0]):
            best = (us, steps)
    t = s.trace()
    sig = [int(x) for x in t["steps"]] + [int(x) for x in np.ascontiguousarray(t["bestL"], np.float32).view(np.uint32)]
    print(json.dumps(dict(us=best[0], steps=best[1], sig=sig)))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child(int(args[1]))
    G, rounds = 12, 3
    while args and args[0].startswith("--"):
        if args[0] == "--gens":
            G = int(args[1])
        elif args[0] == "--rounds":
            rounds = int(args[1])
        args = args[2:]
    libs = args
    res = {l: [] for l in libs}
    sig0 = None
    for r in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(G)], env=dict(os.environ, WELDACS_LIB=os.path.abspath(l)),
                                 capture_output=True, text=True)
            if out.returncode != 0:
                print("%s: FAILED\n%s" % (l, out.stderr[-800:]))
                res[l].append(None)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            if sig0 is None:
                sig0 = d["sig"]
            d["same"] = d["sig"] == sig0
            res[l].append(d)
    for l in libs:
        ok = [d for d in res[l] if d]
        if not ok:
            continue
        tot = [sum(d["us"]) for d in ok]
        ns = [1e3 * sum(d["us"]) / sum(d["steps"]) for d in ok]
        print("%-40s walk over %d generations: %s us  (min %.1f)   %.1f ns per step of the longest walks   trace %s" % (
            os.path.basename(l), G, " ".join("%.1f" % t for t in tot), min(tot), min(ns), "== first" if all(d["same"] for d in ok) else "DIFFERS"))


if __name__ == "__main__":
    main()
