#!/usr/bin/env python3
"""The evaporation sweep alone (k_evaporate, ACSRank_3D.hpp:268-272) -- the command the rocprofv3 kernel-trace and PMC
passes under profiles/ are taken from:   python3 tools/sweep_only.py [grid] [launches]
Prints the per-dispatch event average the same way bench.py's `roofline` does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from welding_robot_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = api.Context(0)
r = bench.sweep_roofline(ctx, n, reps)
r["frac_of_8TBps"] = r["achieved"] / bench.HBM_PEAK_GBS
print(json.dumps(r))
