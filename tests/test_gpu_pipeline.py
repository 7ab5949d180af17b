"""Pipelined groups of searches (wa_acs_set_pipeline; ACSRank_3D.hpp:472-499 is a loop over INDEPENDENT searches): the active slots of
a solver advance in groups on HIP streams of their own, one group's evaporation sweep under another group's walk.  Every slot's own
launch order is unchanged, so whatever the split -- one stream, two groups, one group per search -- every search must equal the oracle
run on its own: per-generation trace (best cost, steps, finite ants, colony), the ants of the last generation, the best path, the
whole pheromone field.  Dense searches also hand their stragglers over PER SLOT in launches that carry several searches."""
import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import api
from test_gpu_edges import bits, box_grid

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def case():
    og = box_grid(48, 40, 44, occ_prob=0.1, seed=21)
    n = 48 * 40 * 44
    og.free[0] = og.free[-1] = og.free[n // 2 + 7] = 1
    return og, n


_oracle_cache = {}


def oracle_run(og, start, end, iters, predict, ants, seed, stream, nb=6):
    key = (id(og), start, end, iters, predict, ants, seed, stream, nb)
    if key not in _oracle_cache:
        a = O.Acs(og, nb=nb)
        tr = a.solve(start, end, iters, predict, fixed_colony=ants, mode=O.DEV, seed=seed, stream=stream)
        olens, oL = a.last_ants()
        _oracle_cache[key] = (tr["steps"].copy(), tr["finite"].copy(), bits(tr["bestL"]).copy(), tr["colony"].copy(), bits(oL).copy(), olens.copy(),
                              bits(a.best_L).copy(), a.best_path()[0].copy(), bits(a.pheromone()).copy())
    return _oracle_cache[key]


def slot_state(s, q, iters):
    t = s.trace(q)
    L, lens = s.ants(q)
    cost, path, _ = s.result(q)
    return (t["steps"][:iters], t["finite"][:iters], bits(t["bestL"][:iters]), t["colony"][:iters], bits(L), lens, bits(cost), path, bits(s.pheromone(q)))


def assert_slot(got, want, tag):
    names = ("steps", "finite", "bestL", "colony", "ant L", "ant nodes", "cost", "best path", "field")
    for g, w, nme in zip(got, want, names):
        assert np.array_equal(g, w), (tag, nme)


@pytest.mark.parametrize("kind", ["dense", "lazy", "nb26"])
@pytest.mark.parametrize("P,groups", [(2, 1), (2, 2), (3, 2), (8, 1), (8, 2), (8, 4), (8, 8), (5, 0)])
def test_pipelined_groups_equal_the_oracle(ctx, case, kind, P, groups):
    og, n = case
    nb = 26 if kind == "nb26" else 6
    iters, ants, predict, seed = 12, 64, 132.0, 4711
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, P, ants, neighbourhood=nb, lazy=kind == "lazy")
    s.set_pipeline(groups)
    ends = [n - 1 if q % 3 else n // 2 + 7 for q in range(P)]     # two distinct end points: shared heuristic fields across groups
    streams = [10 + q for q in range(P)]
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=seed)
    out = np.zeros(16, np.uint64)
    s.init_pheromone(1.0)
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
    s.solve(p, [0] * P, ends, streams=streams)
    ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
    if groups:
        assert s.pipeline_groups() == min(groups, P)
    if kind != "lazy":   # per-slot straggler pools: the hand-over engages in launches that carry several searches, and nothing is left over
        assert int(out[9]) > 0 and int(out[9]) == int(out[7])
    for q in range(P):
        assert_slot(slot_state(s, q, iters), oracle_run(og, 0, ends[q], iters, predict, ants, seed, streams[q], nb), (kind, P, groups, q))
    s.close()
    dg.close()


@pytest.mark.parametrize("chunks", [(1, 1, 1, 1, 1, 1), (2, 3, 1), (4, 2)])
def test_pipelined_groups_across_run_boundaries_and_second_batch(ctx, case, chunks):
    """the groups are forked from and joined into the context's stream inside every wa_acs_run: a search run in pieces (results read in
    between), then a second, smaller batch on the same solver (slots that sat out keep their buffers straight)"""
    og, n = case
    P, ants, predict, seed, iters = 4, 48, 132.0, 99, sum(chunks)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, P, ants)
    s.set_pipeline(4)
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=seed)
    s.init_pheromone(1.0)
    s.begin(p, [0] * P, [n - 1] * P, streams=[1, 2, 3, 4])
    done = 0
    for c in chunks:
        s.run(c)
        done += c
        cost, _, _ = s.result(P - 1)                                     # (a read in between: ordered behind every group)
        assert bits(cost) == oracle_run(og, 0, n - 1, iters, predict, ants, seed, 4)[2][done - 1]
    s.sync()
    for q in range(P):
        assert_slot(slot_state(s, q, iters), oracle_run(og, 0, n - 1, iters, predict, ants, seed, q + 1), ("chunks", q))
    # second batch: three searches, an odd number of generations, two groups; no reset in between for slot 0 .. 2 (carry-over is the
    # reference's behaviour when reset() is not called) -- compared with the same thing on one stream
    res = {}
    for g in (1, 2):
        s.set_pipeline(4)
        s.init_pheromone(1.0)
        s.begin(p, [0] * P, [n - 1] * P, streams=[1, 2, 3, 4])
        s.run(3)
        s.set_pipeline(g)
        p2 = api.default_params(max_iteration=5, predict=predict, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=seed + 1)
        s.solve(p2, [0] * 3, [n // 2 + 7] * 3, streams=[7, 8, 9])
        res[g] = [slot_state(s, q, 5) for q in range(3)]
    for q in range(3):
        assert_slot(res[2][q], res[1][q], ("second batch", q))
    s.close()
    dg.close()


def test_groups_by_rule(ctx, case):
    """the library's own choice (wa_acs_set_pipeline(0)): one group for a lone search, two as soon as there are two dense or four lazy searches, three for
    six and more lazily evaporating searches with colonies of 128 ants and more (profiles/r06/groups_queues.txt); results are those of one stream"""
    og, n = case
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    for lazy, ants, P, want in ((False, 48, 1, 1), (False, 48, 2, 2), (False, 128, 8, 2), (True, 48, 3, 1), (True, 48, 8, 2), (True, 128, 5, 2), (True, 128, 8, 3)):
        s = api.AcsSolver(ctx, dg, P, ants, lazy=lazy)
        p = api.default_params(max_iteration=4, predict=132.0, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=11)
        hist = {}
        for groups in (0, 1):
            s.set_pipeline(groups)
            s.init_pheromone(1.0)
            s.solve(p, [0] * P, [n - 1] * P, streams=list(range(20, 20 + P)))
            if groups == 0:
                assert s.pipeline_groups() == want, (lazy, ants, P, s.pipeline_groups())
            hist[groups] = [slot_state(s, q, 4) for q in range(P)]
        for q in range(P):
            assert_slot(hist[0][q], hist[1][q], ("by rule", lazy, ants, P, q))
        s.close()
    dg.close()
