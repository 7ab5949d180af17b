"""A saturated batch gives the same answer every time it is solved: the searches of a batch are independent (ACSRank_3D.hpp:472-499 runs them one after
another) and every kernel's output must be independent of how its blocks are scheduled.  224 lazily evaporating pair searches on 128^3 -- more walk
blocks than the chip holds, the post-walk launch with few, long background blocks (WA_LAZY_BLOCKS=2: what made a late wavefront of the publishing
block 20 x more likely to read the best it had just published, round 6) -- and 32 dense 256-ant searches, several times each, in one and two
pipelined groups; traces, costs and paths must be equal.  (Parity with the oracle is the other tests' business; this one is about scheduling.)"""
import os

import numpy as np
import pytest

from welding_robot_amd import api, synth

pytestmark = pytest.mark.gpu


def solve_repeatedly(ctx, grid, pts, slots, colony, lazy, gens, runs, groups):
    pairs = [(i, j) for i in range(len(pts)) for j in range(i + 1, len(pts))][:slots]
    a, b = [int(pts[i]) for i, _ in pairs], [int(pts[j]) for _, j in pairs]
    p = api.default_params(max_iteration=gens, predict=float(colony / 0.35), rng_mode=api.RNG_DEV, seed=7)
    s = api.AcsSolver(ctx, grid, n_slots=len(pairs), max_colony=colony, lazy=lazy)
    first, differing = None, []
    for rep in range(runs):
        ctx.check(ctx.lib.wa_acs_set_pipeline(s.h, groups[rep % len(groups)]))
        s.init_pheromone(1.0) if rep == 0 else s.reset_pheromone(1.0)
        s.solve(p, a, b, streams=list(range(len(pairs))))
        costs, paths = s.results(len(pairs))
        cur = [(s.trace(q)["steps"].copy(), s.trace(q)["bestL"].view(np.uint32).copy(), np.float32(costs[q]).view(np.uint32), paths[q]) for q in range(len(pairs))]
        if first is None:
            first = cur
        else:
            differing += [(rep, q) for q in range(len(pairs)) if not all(np.array_equal(x, y) for x, y in zip(first[q], cur[q]))]
    s.close()
    return differing


@pytest.mark.timeout(600)
def test_a_saturated_lazy_batch_is_the_same_every_time():
    old = os.environ.get("WA_LAZY_BLOCKS")
    os.environ["WA_LAZY_BLOCKS"] = "2"           # (read when the solver is created)
    ctx = api.Context(0)
    try:
        n = 128
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        pts = synth.synth_weld_points(free, n, 64, seed=7)
        assert solve_repeatedly(ctx, grid, pts, 224, 24, True, 40, 8, (1, 2)) == []
    finally:
        ctx.close()
        if old is None:
            os.environ.pop("WA_LAZY_BLOCKS", None)
        else:
            os.environ["WA_LAZY_BLOCKS"] = old


@pytest.mark.timeout(600)
def test_a_dense_multi_start_batch_is_the_same_every_time():
    ctx = api.Context(0)
    try:
        n = 96
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, 2024, 0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        pts = synth.synth_weld_points(free, n, 12, seed=3)
        assert solve_repeatedly(ctx, grid, pts, 32, 128, False, 30, 5, (1, 2)) == []
    finally:
        ctx.close()
