#!/usr/bin/env python3
"""What the fused post-walk launch (k_evap_rank_mark: sweep + rank + mark) of a lone 128^3 / 256-ant search spends its time on, in the loop:
the launch as built and without its marks (timing aid of the -DWA_TEST_KNOBS build, WA_SWEEP_NT bit 9: rank + publish stay, nothing is deposited,
the colony keeps exploring).      python tools/fused_parts.py [n]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from welding_robot_amd import api, synth, _lib
n = int(sys.argv[1])
ctx = api.Context(0, lib_path=os.path.join(os.path.dirname(_lib.LIB_PATH), "libweldacs_knobs.so"))
free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=256)
for gens, tag in ((25, "generations 5..24 (exploring)"), (300, "generations 100..299 (converged)")):
    p = api.default_params(max_iteration=gens, predict=731.43 * n / 128, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.init_pheromone(1.0)
    s.begin(p, ids[0], ids[1], streams=[0])
    first = 5 if gens == 25 else 100
    s.run(first); s.sync()
    s.profile(True, 1)
    t0 = time.perf_counter(); s.run(gens - first); s.sync(); dt = time.perf_counter() - t0
    pr = s.profile_read()
    s.profile(False, 1)
    print("  %%-34s walk %%6.1f  fused %%6.2f  apply+table %%5.1f us per launch (every launch stamped)" %% (tag, *[pr[k]["ms"] / max(pr[k]["launches"], 1) * 1e3 for k in ("walk", "evaporate", "deposit")]))
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "128"
# (bit 8, the launch without its sweep, leaves the destination buffer unwritten: the next walk then reads whatever the buffer held -- a timing aid for
#  a single stamped launch under a debugger, not for a running search; not used here)
for nt, what in (("0", "as built"), ("512", "without the marks")):
    env = dict(os.environ, WA_SWEEP_NT=nt)
    print("k_evap_rank_mark %s (WA_SWEEP_NT=%s):" % (what, nt), flush=True)
    subprocess.run([sys.executable, "-c", CHILD, n], env=env)
