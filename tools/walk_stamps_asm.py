#!/usr/bin/env python3
"""Where does a step of the hand-scheduled walk loop spend its cycles?  Diagnostic only (-DWA_ASM_STAMPS build).
Each stamp drains LDS and costs ~40 cycles: read the SHARES.   python tools/walk_stamps_asm.py [generations]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "build", "variants", "stamps_asm.so")
NAMES = ["head: addresses + record loads", "wait for the tabu probe (lgkmcnt)", "compares + wait for the records (vmcnt)",
         "masks, collision branch, touch loads, ordered sums", "draw, readlane, compare, rare-event branch", "pick, insert, next probe, bookkeeping"]
if os.environ.get("WELDACS_LIB") != VARIANT:
    if not os.path.exists(VARIANT) or "--build" in sys.argv:
        from welding_robot_amd import build
        os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
        build.build(out=VARIANT, extra=["-DWA_ASM_STAMPS"])
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--build"], env=dict(os.environ, WELDACS_LIB=VARIANT)))
import numpy as np
from welding_robot_amd import api, synth
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = api.Context(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(128, 2024, 0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
s = api.AcsSolver(ctx, grid, 1, 256)
p = api.default_params(max_iteration=gens, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
s.profile(True, 1)
s.solve(p, 16513, 2097151)
out = np.zeros(16, np.uint64)
ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
steps = int(out[8]); tot = float(out[:6].sum())
print("ant 0: %d steps over %d generations, %.0f stamped cycles/step" % (steps, gens, tot / max(steps, 1)))
for i in range(6):
    print("  %-58s %7.1f /step  %5.1f %%" % (NAMES[i], out[i] / max(steps, 1), 100.0 * out[i] / tot))
pr = s.profile_read()
print("walk kernel (stamped build): %.1f us/generation" % (1e3 * pr["walk"]["ms"] / pr["walk"]["launches"]))
