#!/usr/bin/env python3
"""Many big solvers of ever-changing shape on one context (round 5's arena, csrc/weldacs.hip): every new shape is built from the chunks the
previous ones gave back, in a fresh address range (ranges are never mapped twice: tools/ubench/vmm_reuse.hip) -- so a long-lived process
walks through the 64 TiB the arena takes its ranges from.  This runs N reshapes of a C5-sized lazy solver (random slot counts), a short
search on each, and reports creation times, what was created fresh, and what happens when the address window is used up (the arena then
stops building blocks and the allocator falls back to whole hipMalloc blocks).

    python tools/arena_stress.py [reshapes] [grid]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from welding_robot_amd import api, synth  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ctx = api.Context(0)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(free, n, 8, seed=7)
    p = api.default_params(max_iteration=3, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
    rs = np.random.RandomState(1)
    top, _ = api.pair_slots_by_rule(ctx, grid, 24, 2016, 63, 150, lazy=True)
    want = None
    times, fresh = [], []
    t_all = time.perf_counter()
    for i in range(N):
        slots = int(rs.randint(max(8, top - 60), top + 1))
        before = ctx.cache_stats()
        t0 = time.perf_counter()
        s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        after = ctx.cache_stats()
        s.solve(p, [int(pts[0]), int(pts[2])], [int(pts[1]), int(pts[3])], streams=[0, 1])
        c = [float(s.result(q)[0]) for q in range(2)]
        if want is None:
            want = c
        assert c == want, (i, c, want)           # the same two searches give the same costs on every solver, wherever its memory came from
        s.close()
        times.append(dt)
        fresh.append((after["miss_bytes"] - before["miss_bytes"]) / 2 ** 30)
        if i % 50 == 49 or i == N - 1:
            free_b, total = ctx.memory_info()
            print("reshapes %4d-%4d: creation %.3f s median, %.3f s slowest; %.1f GiB created fresh in these; kept %.1f GiB; out-of-memory events %d; device free (incl. kept) %.1f of %.1f GiB"
                  % (i - 49 if i >= 49 else 0, i, float(np.median(times[-50:])), max(times[-50:]), sum(fresh[-50:]), ctx.cached_bytes() / 2 ** 30,
                     after["oom_events"], free_b / 2 ** 30, total / 2 ** 30), flush=True)
    print("%d reshapes of a %d^3 lazy solver with %d..%d slots in %.1f s; every pair of searches gave the same costs" % (N, n, max(8, top - 60), top, time.perf_counter() - t_all))
    ctx.trim()
    free_b, total = ctx.memory_info()
    print("after wa_ctx_trim: kept %.1f GiB, device free %.1f of %.1f GiB" % (ctx.cached_bytes() / 2 ** 30, free_b / 2 ** 30, total / 2 ** 30))


if __name__ == "__main__":
    main()
