"""Debug aid: first generation / ant / node where the device walk differs from the oracle on the C3 workload (DEV mode)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from welding_robot_amd import api
import oracle_lib as O

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
maxg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = api.Context(0)
og = O.synth_grid(128, seed=2024, occ_prob=0.10)
dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
def both(g):
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=256)
    p = api.default_params(max_iteration=g, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.init_pheromone(1.0)
    s.solve(p, 16513, 2097151)
    a = O.Acs(og)
    tr = a.solve(16513, 2097151, g, 731.43, fixed_colony=256, mode=O.DEV, seed=12345, stream=0)
    return s, a, s.trace(), tr
prev_best = None
for g in range(1, maxg + 1):
    s, a, t, tr = both(g)
    L, lens = s.ants(); olens, oL = a.last_ants()
    paths = a.last_paths()
    nbad = 0
    for i in range(len(lens)):
        dp, op = s.ant_path(i), paths[i]
        if len(dp) != len(op) or (dp != op).any():
            m = min(len(dp), len(op))
            k = int(np.argmax(dp[:m] != op[:m])) if (dp[:m] != op[:m]).any() else m
            if nbad < 8:
                onb = np.isin(op[max(0, k - 2):k + 2], prev_best) if prev_best is not None else None
                print("gen", g, "ant", i, "first differing node index", k, "len dev/ora", len(dp), len(op), "L", L[i], oL[i], "dev", dp[max(0, k - 3):k + 3], "ora", op[max(0, k - 3):k + 3], "on best", onb)
            if nbad == 0:
                neq = np.nonzero(dp[:m] != op[:m])[0]
                print('   differing indices:', neq[:8], '...', neq[-8:], 'count', len(neq), 'zeros', int((dp == 0).sum()))
            nbad += 1
    fd = int((bits(s.pheromone()) != bits(a.pheromone())).sum())
    print("gen", g, "ants with a different path:", nbad, "field words differing:", fd, "best len", len(a.best_path()[0]))
    prev_best = a.best_path()[0].copy()
    s.close()
    if nbad or fd: break
