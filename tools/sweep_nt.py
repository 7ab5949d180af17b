#!/usr/bin/env python3
"""The sweep kernel alone (k_evaporate, 64 stamped launches) at 128^3 and 256^3 under every cache policy (WA_SWEEP_NT=0..3: bit 0
non-temporal loads, bit 1 non-temporal stores); the library's rule picks 0 below the Infinity Cache and 3 past it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import bench
    from welding_robot_amd import api
    ctx = api.Context(0)
    for n in (128, 256):
        r = bench.sweep_roofline(ctx, n)
        print("WA_SWEEP_NT=%s  %d^3: %.2f us per launch, %.0f GB/s = %.3f of 8 TB/s" % (os.environ.get("WA_SWEEP_NT", "rule"), n, r["avg_launch_ms"] * 1e3, r["achieved"], r["achieved"] / 8000.0))
else:
    for nt in ("rule", "0", "1", "2", "3"):
        env = dict(os.environ)
        if nt != "rule":
            env["WA_SWEEP_NT"] = nt
        else:
            env.pop("WA_SWEEP_NT", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
