#!/usr/bin/env python3
"""Anatomy of the walk launches with the straggler hand-over (diagnostic -DWA_STRAG_TIME build: wall-clock stamps per generation):
when is the 51st arrival published, when do the last arriving / handed-over / resumed ants' blocks end, and how many nodes had the
last ones -- against the node count of the 51st shortest ant (a second solver without the hand-over, stepped one generation at a time).
    python tools/straggler_time.py --build      (here: cross-compiles build/variants/strag_time.so)
    python tools/straggler_time.py              (on the GPU box)"""
import os, subprocess, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "build", "variants", "strag_time.so")
if "--build" in sys.argv:
    from welding_robot_amd import build
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    print(build.build(out=VARIANT, extra=["-DWA_STRAG_TIME"] + [a for a in sys.argv[1:] if a.startswith("-D")]))
    sys.exit(0)
if os.environ.get("WELDACS_LIB") != VARIANT:
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, WELDACS_LIB=VARIANT)))
import numpy as np
from welding_robot_amd import api, synth
from welding_robot_amd import dist as wd
n, ants = 128, 256
ctx = api.Context(0)
wl = wd.per_rank_workload(0)
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=wl["grid_seed"], occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants)
p = api.default_params(max_iteration=40, predict=3.0 * n, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=wl["rng_seed"])
out = np.zeros(128 * 8, np.uint64)
rd = ctx.lib.wa_strag_time_read; rd.argtypes = [ctypes.c_void_p, ctypes.c_int32]
for rep in range(2):
    s.init_pheromone(1.0)
    s.begin(p, ids[0], ids[1], streams=[wl["stream"]])
    rd(out.ctypes.data, 1)
    s.run(40); s.sync()
rd(out.ctypes.data, 0)
os.environ["WA_STRAGGLERS"] = "0"
s2 = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants)
s2.init_pheromone(1.0); s2.begin(p, ids[0], ids[1], streams=[wl["stream"]])
lens_by_gen = []
for g in range(40):
    s2.run(1); s2.sync()
    L, ln = s2.ants()
    lens_by_gen.append(np.where(np.isfinite(L), ln, 0))
t = out.reshape(128, 8)
print("# 100 MHz wall clock, us since the first block of the walk launch started: 51st arrival published | last arrived ant's block ends | last handed-over / dead ant's block ends | last resume block ends | arrived ants")
tot = np.zeros(4)
for g in range(40):
    t0 = (~t[g, 0]) & np.uint64(0xffffffffffffffff)
    f = lambda x: (float(int(x) - int(t0)) / 100.0) if x else float('nan')
    row = [f(t[g, 1]), f(t[g, 2]), f(t[g, 4]), f(t[g, 3])]
    la, lh = int(t[g, 6]) & 0xffff, int(t[g, 7]) & 0xffff
    L = lens_by_gen[g]; fin = np.sort(L[L > 0]) if len(L) else []
    l51 = int(fin[50]) if len(fin) > 50 else -1
    print("gen %2d   %7.1f %7.1f %7.1f %7.1f   %3d   nodes of the last arrived ant %4d, of the last handed-over ant at its hand-over %4d; 51st shortest ant %4d" % (g, *row, int(t[g, 5]), la, lh, l51))
