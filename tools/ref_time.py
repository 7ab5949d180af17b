#!/usr/bin/env python3
"""REF mode (the reference's own libc stream, ants one after another on one wavefront) at C3: seconds for N generations of the 128^3 /
256-ant search, with the hand-scheduled loop (default) and with the compiler-scheduled one (WA_WALK_ASM=0: what REF walks ran on
before round 4)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from welding_robot_amd import api, synth
    gens = int(sys.argv[2])
    ctx = api.Context(0)
    n = 128
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=256)
    p = api.default_params(max_iteration=gens, predict=731.43, fixed_colony=256, rng_mode=api.RNG_REF)
    s.srand(12345)
    s.init_pheromone(1.0)
    t0 = time.perf_counter()
    s.solve(p, ids[0], ids[1])
    dt = time.perf_counter() - t0
    tr = s.trace()
    print("WA_WALK_ASM=%s  %d REF generations: %.2f s (%.1f generations/s), %d steps in all = %.0f ns per step, best cost %.1f" % (
        os.environ.get("WA_WALK_ASM", "1"), gens, dt, gens / dt, int(tr["steps"].sum()), dt * 1e9 / tr["steps"].sum(), s.result()[0]))
else:
    gens = sys.argv[1] if len(sys.argv) > 1 else "500"
    for asm in ("1", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", gens], env=dict(os.environ, WA_WALK_ASM=asm))
