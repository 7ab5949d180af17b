#!/usr/bin/env python3
"""Where the time of wa_acs_create goes at BASELINE config C5's shape (256^3, lazy, 24 ants, slots by rule): raw hipMalloc / hipMemset /
hipFree of the same bytes through the HIP runtime, then the solver itself -- created in a fresh process, destroyed, created again.

    python tools/create_time.py [--raw] [grid] [slots ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from welding_robot_amd import api, synth  # noqa: E402


def raw(hip, total, chunk):
    ptrs = []
    t0 = time.perf_counter()
    left = total
    while left > 0:
        p = C.c_void_p()
        b = min(chunk, left)
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(b))
        if rc:
            print("   hipMalloc rc", rc)
            break
        ptrs.append((p, b))
        left -= b
    t1 = time.perf_counter()
    for p, b in ptrs:
        hip.hipMemsetAsync(p, 0, C.c_size_t(b), None)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    for p, b in ptrs:
        hip.hipMemsetAsync(p, 0, C.c_size_t(b), None)
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    for p, _ in ptrs:
        hip.hipFree(p)
    t4 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t4 - t3


def main():
    do_raw = "--raw" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--raw"]
    n = int(args[0]) if args else 256
    slot_list = [int(a) for a in args[1:]] or [224, 96]
    ctx = api.Context(0)
    hip = C.CDLL("libamdhip64.so")
    free, total = ctx.memory_info()
    print("device memory: %.1f GB free of %.1f" % (free / 1e9, total / 1e9))
    if do_raw:
        for gb, chunk in ((64, 1 << 30), (64, 1 << 30), (190, 1 << 30), (190, 1 << 30), (190, 400 << 20)):
            a, b, c, d = raw(hip, int(gb * 1e9), chunk)
            print("raw %3d GB in %5d MiB blocks: hipMalloc %.3f s, first memset %.3f s, second memset %.3f s, hipFree %.3f s" % (gb, chunk >> 20, a, b, c, d))
        t0 = time.perf_counter()
        while ctx.memory_info()[0] < 0.95 * total and time.perf_counter() - t0 < 60:
            time.sleep(0.05)
        print("memory back after %.2f s" % (time.perf_counter() - t0))
    free_np, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free_np, cx, cy, cz, prec, wall)
    per_slot, per_field, fixed = api.memory_estimate(grid, 24, 0, 6, True)
    print("estimate: %.3f GB per slot, %.3f GB per heuristic field" % (per_slot / 1e9, per_field / 1e9))
    for slots in slot_list:
        for rep in range(3):
            t0 = time.perf_counter()
            s = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=24, lazy=True)
            ctx.sync()
            t1 = time.perf_counter()
            s.close()
            ctx.sync()
            t2 = time.perf_counter()
            print("solver %3d slots (%.0f GB), round %d: create %.3f s, destroy %.3f s" % (slots, slots * per_slot / 1e9, rep, t1 - t0, t2 - t1))
    ctx.close()


if __name__ == "__main__":
    main()
