#!/usr/bin/env python3
"""The drop-in C++ classes (examples/dropin_demo.cpp = main.cpp:273-352 on libweldacs) at BASELINE config C5's scale: origin_piece.stl (the
reference's largest mesh, 29 888 triangles) voxelised finely, 64 weld points on free voxels, all 2 016 pair searches x 150 generations
sized and dealt by the header's own rules, seam order, stitching, two smoothing passes.  Wall time of the whole process and of its parts.

    python tools/dropin_c5.py [precision] [points]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from welding_robot_amd import api  # noqa: E402
import test_dropin as TD  # noqa: E402


def main():
    prec = float(sys.argv[1]) if len(sys.argv) > 1 else 0.004
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    stl = os.path.join(ROOT, "tests", "golden", "origin_piece.stl")
    ctx = api.Context(0)
    g = api.Grid.from_mesh(ctx, api.stl_read_file(stl), prec, 4)
    occ = g.occupancy()
    cx, cy, cz = g.coords()
    rs = np.random.RandomState(5)
    fr = np.flatnonzero(occ)
    ids = rs.choice(fr, P, replace=False)
    pts = np.stack([cx[ids % g.nx], cy[(ids // g.nx) % g.ny], cz[ids // (g.nx * g.ny)]], 1)
    dims = (g.nx, g.ny, g.nz)
    g.close(); ctx.close()
    pf = "/tmp/weldacs_c5_points.in"
    with open(pf, "w") as f:
        f.write("%d\n" % P)
        for p in pts:
            f.write("%.9g %.9g %.9g\n" % tuple(p))
    exe = TD.compile_demo()
    out = "/tmp/weldacs_c5_dropin.txt"
    t0 = time.time()
    os.environ.setdefault("WA_DEMO_TIMES", "1")
    os.environ.setdefault("WA_TRACE_CREATE", "1")
    r = subprocess.run([exe, stl, repr(prec), "4", pf, str(24 / 0.35 * prec), out + ".graph", "dev", "7", out], capture_output=True, text=True)
    dt = time.time() - t0
    print("grid %dx%dx%d = %.1f M voxels, %d points = %d pair searches x 150 generations (24 ants): process %.2f s, rc %d" % (
        dims + (dims[0] * dims[1] * dims[2] / 1e6, P, P * (P - 1) // 2, dt, r.returncode)))
    for l in r.stderr.splitlines():
        if l.startswith("[demo]") or l.startswith("[weldacs]"):
            print("   " + l.strip()[:200])
    for l in r.stdout.splitlines():
        if "shard" in l.lower() or "slots" in l.lower() or "[ACS 3D] Created" in l or "time" in l.lower():
            print("   " + l.strip()[:200])
    d = TD.parse(out)
    print("   tour_L %.3f, %d iterations, stitched path %d nodes, smooth2 %d samples" % (d["tour_L"], d["iters"], len(d["gpath"]), len(d["smooth2"])))


if __name__ == "__main__":
    main()
