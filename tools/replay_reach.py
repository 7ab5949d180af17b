#!/usr/bin/env python3
"""How far do ants follow the best path before their first deviation, generation by generation?  Diagnostic
(-DWA_STAMPS build): tells how much of the pre-convergence walk a replay of ONE path can cover."""
import os, subprocess, sys
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = "/tmp/libweldacs_stamps.so"
if os.environ.get("WELDACS_LIB") != VARIANT:
    from welding_robot_amd import build
    build.build(out=VARIANT, extra=["-DWA_STAMPS"])
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, WELDACS_LIB=VARIANT)))
import numpy as np
from welding_robot_amd import api, synth
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
s = api.AcsSolver(ctx, grid, 1, 256)
p = api.default_params(max_iteration=120, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
s.begin(p, 16513, 2097151)
out = np.zeros(16, np.uint64)
prev = np.zeros(16, np.uint64)
rows = []
for g in range(120):
    s.run(1); s.sync()
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
    d = out.astype(np.int64) - prev.astype(np.int64); prev = out.copy()
    tr = s.trace()
    rows.append((g, int(d[11]), d[10] / max(d[11], 1), int(d[12]), tr["steps"][g] / 256.0, float(tr["bestL"][g])))
for lo, hi in ((0, 10), (10, 20), (20, 30), (30, 40), (40, 50), (50, 60), (60, 70), (70, 80), (80, 90), (90, 120)):
    r = [x for x in rows if lo <= x[0] < hi]
    print("gens %3d-%3d: ants replaying %5.1f /256, nodes followed before the first deviation %6.1f, arrived on the replay track %5.1f /256, steps/ant %6.1f, best %g" % (
        lo, hi, np.mean([x[1] for x in r]), np.mean([x[2] for x in r]), np.mean([x[3] for x in r]), np.mean([x[4] for x in r]), r[-1][5]))
