#!/usr/bin/env python3
"""Where does a step of the walk spend its cycles?  Diagnostic only.

Builds libweldacs with -DWA_STAMPS into a scratch file, runs the bench workload (128^3, 256 ants)
for a number of generations and prints the share of each inner-loop section for ant 0, plus the
product build's plain walk time for reference.  Stamps perturb the loop (each drains LDS and costs
~40 cycles): read the SHARES, not the totals.

    python tools/walk_stamps.py [generations]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = "/tmp/libweldacs_stamps.so"
NAMES = ["back-edge + record wait", "prefetch issue", "probe wait + collisions", "admissible + scans", "draw + pick",
         "insert + next probe", "path capture + counters", "-"]


def main():
    gens = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    if os.environ.get("WELDACS_LIB") != VARIANT:
        from welding_robot_amd import build
        build.build(out=VARIANT, extra=["-DWA_STAMPS"])
        env = dict(os.environ, WELDACS_LIB=VARIANT)
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    import numpy as np
    from welding_robot_amd import api, synth
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    s = api.AcsSolver(ctx, grid, 1, 256)
    p = api.default_params(max_iteration=gens, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.profile(True, 1)
    s.solve(p, 16513, 2097151)
    out = np.zeros(16, np.uint64)
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
    steps = int(out[8])
    tot = float(out[:7].sum())
    print("ant 0: %d steps over %d generations, %.0f stamped cycles/step (s_memtime = 100 MHz ticks? see below)" % (steps, gens, tot / max(steps, 1)))
    for i in range(7):
        print("  %-28s %8.1f /step  %5.1f %%" % (NAMES[i], out[i] / max(steps, 1), 100.0 * out[i] / tot))
    pr = s.profile_read()
    print("walk kernel (stamped build): %.1f us/generation" % (1e3 * pr["walk"]["ms"] / pr["walk"]["launches"]))


if __name__ == "__main__":
    main()
