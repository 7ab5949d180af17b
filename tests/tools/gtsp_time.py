import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from welding_robot_amd import api
import oracle_lib as O
ctx = api.Context(0)
rs = np.random.RandomState(4242)
for n in (16, 64, 128, 256):
    P = rs.randint(0, 100, size=(n, 3))
    d = np.abs(P[:, None, :] - P[None, :, :]).sum(-1) / 100.0 + 0.001
    np.fill_diagonal(d, 0)
    api.gtsp_solve(ctx, d, mode=api.RNG_DEV, seed=5)
    t0 = time.perf_counter(); t = api.gtsp_solve(ctx, d, mode=api.RNG_DEV, seed=5); t1 = time.perf_counter()
    line = "n=%d gpu %.2f ms, %d iters, %.1f us/iter, L=%.3f" % (n, (t1 - t0) * 1e3, t["iters"][0], (t1 - t0) * 1e6 / t["iters"][0], t["L"][0])
    if n <= 128:
        t0 = time.perf_counter(); o = O.gtsp_solve(d, mode=O.DEV, seed=5); t1 = time.perf_counter()
        line += " | cpu port %.1f ms, equal=%s" % ((t1 - t0) * 1e3, o["L"] == t["L"][0] and np.array_equal(o["edges"], t["edges"][0]))
    print(line)
# batch of 64 instances
d64 = np.stack([d[:64, :64]] * 64)
t0 = time.perf_counter(); t = api.gtsp_solve(ctx, d64, mode=api.RNG_DEV, seed=5); t1 = time.perf_counter()
print("64 instances of n=64: %.2f ms total" % ((t1 - t0) * 1e3))
