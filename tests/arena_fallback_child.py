#!/usr/bin/env python3
"""Child process of tests/test_gpu_arena.py::test_arena_fallbacks_*: the arena's fall-backs are process-wide states (the address window,
the hint knob), so each runs in a process of its own.

    python tests/arena_fallback_child.py <mode> <c5_slots|0>      mode: noarena | nohint | smallwindow

noarena      WA_DEV_ARENA=0: whole hipMalloc blocks, exact-fit reuse (what a device without virtual memory management runs)
nohint       WA_DEV_ARENA_NOHINT=1: address ranges are reserved WITHOUT hints, i.e. a platform that does not place a range where it is asked
             to -- va_reserve_fresh_locked gives up after eight answers elsewhere, the arena stops building blocks for good
smallwindow  WA_DEV_ARENA_VA_MB=6144: the address window is used up in the middle of the second solver
The sequence is VERDICT r04's: an 8- and a 32-slot dense 128^3 solver, then the big lazy 256^3 one, twice; one pair of it against the
oracle (cost, path, whole field).  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mode, c5_slots = sys.argv[1], int(sys.argv[2])
os.environ["WA_DEV_POISON"] = "1"
if mode == "noarena":
    os.environ["WA_DEV_ARENA"] = "0"
elif mode == "nohint":
    os.environ["WA_DEV_ARENA_NOHINT"] = "1"
elif mode == "smallwindow":
    os.environ["WA_DEV_ARENA_VA_MB"] = "6144"

import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from welding_robot_amd import api, synth  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    ctx = api.Context(0)
    out = {"mode": mode, "arena_at_start": ctx.cache_stats()["arena"]}
    f128, cx, cy, cz, prec, wall = synth.synth_grid(128, seed=2024, occ_prob=0.10)
    g128 = api.Grid.from_occupancy(ctx, f128, cx, cy, cz, prec, wall)
    t_small = []
    for slots in (8, 32):
        t0 = time.perf_counter()
        s = api.AcsSolver(ctx, g128, n_slots=slots, max_colony=256, lazy=False)
        ctx.sync()
        t_small.append(time.perf_counter() - t0)
        s.close()
    f256, cx, cy, cz, prec, wall = synth.synth_grid(256, seed=2024, occ_prob=0.10)
    g256 = api.Grid.from_occupancy(ctx, f256, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(f256, 256, 64, seed=7)
    slots = c5_slots or api.pair_slots_by_rule(ctx, g256, 24, 2016, 63, 150, lazy=True)[0]
    og = O.Grid(cx, cy, cz, f256, prec, wall)
    a = O.Acs(og)
    a.solve(int(pts[2]), int(pts[3]), 20, float(24 / 0.35), mode=O.DEV, seed=7, stream=1)
    p = api.default_params(max_iteration=20, predict=float(24 / 0.35), rng_mode=api.RNG_DEV, seed=7)
    t_big, ok = [], True
    for rep in range(2):
        t0 = time.perf_counter()
        s = api.AcsSolver(ctx, g256, n_slots=slots, max_colony=24, lazy=True)
        ctx.sync()
        t_big.append(time.perf_counter() - t0)
        s.solve(p, [int(pts[0]), int(pts[2])], [int(pts[1]), int(pts[3])], streams=[0, 1])
        cost, path, _ = s.result(1)
        ok = ok and bits(cost) == bits(a.best_L) and (not np.isfinite(cost) or np.array_equal(path, a.best_path()[0]))
        ok = ok and np.array_equal(bits(s.pheromone(1)), bits(a.pheromone()))
        s.close()
    st = ctx.cache_stats()
    out.update(slots=slots, t_small_s=t_small, t_big_s=t_big, equals_oracle=bool(ok), arena_at_end=st["arena"], oom_events=st["oom_events"],
               kept_gib=st["kept_bytes"] / 2.0 ** 30, hit_gib=st["hit_bytes"] / 2.0 ** 30, miss_gib=st["miss_bytes"] / 2.0 ** 30, released_gib=st["released_bytes"] / 2.0 ** 30)
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
