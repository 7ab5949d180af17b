#!/usr/bin/env python3
"""profiles/r04/scalar_calls.txt: microseconds per scalar call of the drop-in (examples/scalar_calls.cpp: getCurvePoint on the host
path and through the device, getCurveDerPoint, setPoints, the sample counts of main.cpp's two clock()-paced loops) plus the C ABI's
result read-backs timed from Python (wa_acs_result, wa_acs_result_batch, wa_grid_resolve_points).

    python tools/scalar_calls.py [out.txt]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from welding_robot_amd import api, synth  # noqa: E402
import test_scalar_calls as T  # noqa: E402


def per_call(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04", "scalar_calls.txt")
    T.compile_exe()
    d = T.run_report("/tmp/weldacs_scalar_report.txt")
    ctx = api.Context(0)
    n = 128
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32)
    ids = grid.resolve(pts)
    s = api.AcsSolver(ctx, grid, n_slots=8, max_colony=256)
    p = api.default_params(max_iteration=60, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.init_pheromone(1.0)
    s.solve(p, [ids[0]] * 8, [ids[1]] * 8, streams=list(range(8)))
    d["wa_grid_resolve_points_2pts_128_us"] = per_call(lambda: grid.resolve(pts), 200)
    d["wa_acs_result_cost_only_us"] = per_call(lambda: ctx.lib.wa_acs_result(s.h, 0, None, None, None, None, 0), 500)
    d["wa_acs_result_with_path_us"] = per_call(lambda: s.result(0), 200)
    d["wa_acs_result_batch_8_slots_us"] = per_call(lambda: s.results(8), 200)
    with open(out, "w") as f:
        f.write("# microseconds per scalar call (tools/scalar_calls.py; examples/scalar_calls.cpp on tests/golden/cubic.stl, the read-backs on a 128^3 grid after a\n"
                "# 60-generation search; the Python figures include ~1-2 us of ctypes marshalling).  loop*_samples: main.cpp:302-316 / :341-351's clock()-paced loops --\n"
                "# 'ideal' = what a call of zero duration would collect; getCurvePoint evaluates single points on the host (wa_bspline_eval_host).\n")
        for k in sorted(d):
            f.write("%-40s %.4f\n" % (k, d[k]))
    print(open(out).read())


if __name__ == "__main__":
    main()
