#!/usr/bin/env python3
"""Per-ant block time against step count (diagnostic -DWA_ANT_TIME build): is a walk launch as long as (its longest walk x the
average step), or are the long walks slower per step / is there a fixed part?   python tools/ant_time.py [generations]"""
import os, subprocess, sys
os.environ.setdefault("WA_STRAGGLER_DRAIN", "0")   # these generation-by-generation measurements assume every ant finishes inside its own launch (round 3 semantics)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "build", "variants", "ant_time.so")
if "--build" in sys.argv:
    from welding_robot_amd import build
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    print(build.build(out=VARIANT, extra=["-DWA_ANT_TIME"] + [a for a in sys.argv[1:] if a.startswith("-D")]))
    sys.exit(0)
if os.environ.get("WELDACS_LIB") != VARIANT:
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, WELDACS_LIB=VARIANT)))
import numpy as np
from welding_robot_amd import api, synth
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 12
show = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
s = api.AcsSolver(ctx, grid, 1, 256)
p = api.default_params(max_iteration=gens, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
for rep in range(2):
    s.init_pheromone(1.0)
    s.begin(p, 16513, 2097151)
    rows = []
    for g in range(gens):
        out = np.zeros(16, np.uint64)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        s.profile(True, 1)
        s.run(1)
        pr = s.profile_read()
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        ph = [int(out[i]) for i in range(5, 11)]
        phases = "entry->init %d | LDS init %d | ->loop %d | loop %d | ->end %d" % (ph[1] - ph[0], ph[2] - ph[1], ph[3] - ph[2], ph[4] - ph[3], ph[5] - ph[4])
        w = int(out[12])
        if int(out[15]):
            phases += "\n        rejoin watch: %d ants, %d hand-backs in all, %.0f ticks per hand-back outside the loop; most hand-backs: %d (gained %d nodes, %d general steps, prefix %d)" % (
                int(out[15]), int(out[14]), int(out[13]) / max(int(out[14]), 1), w >> 48, (w >> 32) & 0xffff, (w >> 16) & 0xffff, w & 0xffff)
        rows.append((phases, g, pr["walk"]["ms"] * 1e3, int(out[1]) >> 24, int(out[1]) & 0xffffff, int(out[4]) >> 32, int(out[4]) & 0xffffffff, int(out[2]), int(out[3])))
for phases, g, us, tslow, nslow, nlong, tlong, tsum, nsum in rows:
    if show is not None and g not in show:
        continue
    print("gen %2d: launch %6.1f us | slowest block %6d ticks (%5.1f us at 2.39 GHz) for %4d steps = %5.1f ticks/step | longest walk %4d steps in %6d ticks = %5.1f/step | all ants %5.1f ticks/step" % (
        g, us, tslow, tslow / 2390.0, nslow, tslow / max(nslow, 1), nlong, tlong, tlong / max(nlong, 1), tsum / max(nsum, 1)))
    print("        ant 0 phases (ticks): " + phases)
