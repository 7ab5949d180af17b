#!/usr/bin/env python3
"""Time the trajectory kernels (stitch, spline setup, batched evaluation) and time the CPU oracle beside them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from welding_robot_amd import api
import oracle_lib as O

ctx = api.Context(0)
rs = np.random.RandomState(5)


def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


# stitched path of a C5-sized tour: 63 segments x ~600 nodes on a 256^3 lattice
n_seg, seg_len, N = 63, 600, 256
cx = np.arange(N, dtype=np.float32)
free = np.ones(N ** 3, np.uint8)
grid = api.Grid.from_occupancy(ctx, free, cx, cx, cx, 1.0, 0)
segs = [rs.randint(0, N ** 3, size=seg_len).astype(np.int64) for _ in range(n_seg)]
t = timed(lambda: api.Trajectory.stitch(grid, segs).close())
print("stitch %d segments x %d nodes (upload ids + gather + free): %.3f ms" % (n_seg, seg_len, t * 1e3))
path = api.Trajectory.stitch(grid, segs)
n = len(path)
xyz = path.points()

for deg, ci, cf, tf, count in [(0, 0, 0, 150.0, 1 << 20), (2, 2, 2, 6000.0, 1 << 20), (3, 2, 2, 6000.0, 1 << 20), (3, 2, 2, 6000.0, 1 << 24)]:
    b = api.Bspline(ctx, 3, deg, ci, cf, n)
    init = np.zeros((ci + 1, 3), np.float32); init[0] = xyz[0]
    fin = np.zeros((cf + 1, 3), np.float32); fin[0] = xyz[-1]
    ts = timed(lambda: b.set_param(init, fin, path, tf))
    dt = np.float32(tf / count)
    te = timed(lambda: b.sample(0.0, dt, count, host=False, device=True)[2].close())
    ob = O.Bspline(3, deg, ci, cf, n)
    t0 = time.perf_counter(); ob.set_param(init, fin, xyz, tf); tos = time.perf_counter() - t0
    cs = min(count, 1 << 18)
    t0 = time.perf_counter(); want, _ = ob.sample(0.0, dt, cs); toe = time.perf_counter() - t0
    got, _ = b.sample(0.0, dt, cs)
    print("BS_Basic<3,%d,%d,%d> %d control points: setup %.3f ms (CPU port %.3f ms); %d samples -> device polyline %.3f ms = %.2f Gsamples/s, "
          "%.1f GB/s written (CPU port %.1f Msamples/s, 1 thread); first %d samples bit-equal: %s" % (
              deg, ci, cf, n + 2 + ci + cf, ts * 1e3, tos * 1e3, count, te * 1e3, count / te / 1e9, count * 12 / te / 1e9, cs / toe / 1e6, cs,
              bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))))
    b.close()
