#!/usr/bin/env python3
"""Cycles per section of one step of the hand-scheduled walk loop, two un-waited s_memtime per step (diagnostic builds
-DWA_ASM_SPAN_A=a -DWA_ASM_SPAN_B=b, see walk_loop_gfx950.hpp).  One build per section.

    python tools/walk_spans.py --build [-DWA_EXP=n]     # here (no GPU needed): build/spans/*.so
    python tools/walk_spans.py [generations]            # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIR = os.path.join(ROOT, "build", "spans")
SPANS = [(1, 2, "record wait (2 compares + s_waitcnt vmcnt)"), (2, 3, "sign compare, info, two touch loads"), (3, 4, "masks (+ collision branch), v_cndmask"),
         (4, 5, "draw readlane, path words, the ten ordered adds"), (5, 6, "rnd, readlane, compare, s_and, rare-event branch"),
         (6, 7, "s_ff1, pick lane, path word readlane, insert"), (7, 8, "active mask, cur, first half of {record loads | probe}"),
         (8, 9, "second half of {record loads | probe}"), (9, 10, "path word, m0, block / arrival events"), (10, 0, "back edge + touch address"), (0, 1, "touch address + s_waitcnt lgkmcnt (the probe's LDS round trip)"),
         (2, 2, "WHOLE STEP (point 2 to point 2)")]


def lib(a, b):
    return os.path.join(DIR, "span_%d_%d.so" % (a, b))


if "--build" in sys.argv:
    from welding_robot_amd import build
    os.makedirs(DIR, exist_ok=True)
    extra = [x for x in sys.argv[1:] if x.startswith("-D")]
    for a, b, _ in SPANS:
        build.build(out=lib(a, b), extra=["-DWA_ASM_SPAN_A=%d" % a, "-DWA_ASM_SPAN_B=%d" % b] + extra)
        print(lib(a, b))
    sys.exit(0)
if os.environ.get("WA_SPAN_CHILD"):
    import numpy as np
    from welding_robot_amd import api, synth
    gens = int(sys.argv[1])
    ctx = api.Context(0)
    n = int(os.environ.get("WA_SPAN_GRID", "128"))   # 32 / 64: the whole field lives in the L2s
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    s = api.AcsSolver(ctx, grid, 1, 256)
    p = api.default_params(max_iteration=gens, predict=731.43 * n / 128, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.profile(True, 1)
    ids = (16513, 2097151) if n == 128 else tuple(int(v) for v in grid.resolve(np.array([[1, 1, 1], [n - 1, n - 1, n - 1]], np.float32)))
    s.solve(p, ids[0], ids[1])
    out = np.zeros(16, np.uint64)
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
    pr = s.profile_read()
    print("%d %d %.2f %d %d" % (int(out[0]), int(out[8]), 1e3 * pr["walk"]["ms"] / pr["walk"]["launches"], int(out[1]), int(out[2])), flush=True)
    s.close(); grid.close(); ctx.close()
    os._exit(0)   # (the HIP runtime's static destructors occasionally throw at interpreter exit)
gens = sys.argv[1] if len(sys.argv) > 1 else "10"
tot = 0.0
if os.environ.get("WA_SPAN_ONLY"):   # e.g. WA_SPAN_ONLY=1-2,2-2
    want = {tuple(int(x) for x in t.split("-")) for t in os.environ["WA_SPAN_ONLY"].split(",")}
    SPANS = [x for x in SPANS if (x[0], x[1]) in want]
for a, b, name in SPANS:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), gens], env=dict(os.environ, WELDACS_LIB=lib(a, b), WA_SPAN_CHILD="1"), capture_output=True, text=True)
    if r.returncode and len(r.stdout.split()) < 5:
        print("span %d-%d failed: %s" % (a, b, r.stderr[-400:]))
        continue
    cyc, steps, us, coll, events = r.stdout.split()[-5:]
    per = float(cyc) / max(int(steps), 1)
    if a != b:
        tot += per
    print("  %2d -> %2d  %-62s %7.1f cycles/step   (walk launch of this build %s us)" % (a, b, name, per, us))
print("  ant 0: %s steps, %s probe collisions (step evaluated again), %s rare events (block boundary / arrival / dead end)" % (steps, coll, events))
print("  sum of the sections %.1f cycles (the LDS wait + span bookkeeping at the head is the whole step minus this)" % tot)
