"""Randomised differential test of the DEV search against the C oracle (bit-exact): random box grids, obstacles, end points, colony
rules and parameters -- the combinations nobody wrote a case for.  Every case is a fixed seed, so a failure reproduces."""
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


def draw_case(seed):
    rs = np.random.RandomState(1000 + seed)
    nx, ny, nz = (int(rs.randint(3, 25)) for _ in range(3))
    p = float(rs.choice([1.0, 0.5, 0.013]))
    og = box_grid(nx, ny, nz, occ_prob=float(rs.choice([0.0, 0.1, 0.2, 0.35])), seed=seed, p=p)
    free = np.flatnonzero(og.free)
    if len(free) < 2:
        og = box_grid(nx, ny, nz, 0.0, seed, p)
        free = np.flatnonzero(og.free)
    sid, eid = (int(v) for v in rs.choice(free, 2, replace=False))
    par = dict(alpha=int(rs.choice([1, 1, 1, 2])), beta=float(rs.choice([0.6, 1.0, 2.5])), rho=float(rs.choice([0.5, 0.8, 0.95])),
               pheromone_0=float(rs.choice([1.0, 0.3])))
    diam = (nx + ny + nz) * p
    if rs.rand() < 0.5:
        fixed, predict = int(rs.randint(1, 97)), diam
    else:   # adaptive colony (:247): 0.35 * min(best, predict) / precision ants, at least a few
        fixed, predict = 0, diam * float(rs.uniform(0.4, 3.0))
        if int(0.35 * predict / p) < 1:
            fixed = 3
    iters = int(rs.randint(1, 41))
    return og, sid, eid, iters, predict, fixed, int(rs.randint(1, 1 << 30)), int(rs.randint(0, 8)), par


def run_variant(ctx, og, sid, eid, iters, predict, fixed, seed, stream, par, nb=6, lazy=False):
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    bound = fixed if fixed else int(0.35 * predict / float(og.precision))
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=max(bound, 1), neighbourhood=nb, lazy=lazy)
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_DEV, seed=seed, **par)
    s.init_pheromone(par["pheromone_0"])
    s.solve(p, sid, eid, streams=[stream])
    a = O.Acs(og, nb=nb, pheromone_0=par["pheromone_0"])
    tr = a.solve(sid, eid, iters, predict, fixed_colony=fixed, mode=O.DEV, seed=seed, stream=stream, alpha=par["alpha"], beta=par["beta"],
                 rho=par["rho"], pheromone_0=par["pheromone_0"])
    t = s.trace()
    assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(t["colony"], tr["colony"]) and np.array_equal(t["finite"], tr["finite"])
    assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(bits(t["iterbestL"]), bits(tr["iterbestL"]))
    cost, path, ch = s.result()
    assert bits(cost) == bits(a.best_L)
    if np.isfinite(cost):
        assert np.array_equal(path, a.best_path()[0]) and np.array_equal(ch.astype(np.int32), a.best_path()[1])
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))          # the whole field
    L, lens = s.ants()
    olen, oL = a.last_ants()
    assert np.array_equal(bits(L), bits(oL)) and np.array_equal(lens, olen)   # every ant of the last generation
    s.close()
    dg.close()


@pytest.mark.parametrize("seed", range(96))
def test_random_search_equals_the_oracle(ctx, seed):
    og, sid, eid, iters, predict, fixed, rng_seed, stream, par = draw_case(seed)
    run_variant(ctx, og, sid, eid, iters, predict, fixed, rng_seed, stream, par)


@pytest.mark.parametrize("seed", range(100, 132))
def test_random_search_with_lazy_evaporation_equals_the_oracle(ctx, seed):
    og, sid, eid, iters, predict, fixed, rng_seed, stream, par = draw_case(seed)
    run_variant(ctx, og, sid, eid, iters, predict, fixed, rng_seed, stream, par, lazy=True)


@pytest.mark.parametrize("seed", range(200, 232))
def test_random_26_neighbour_search_equals_the_oracle(ctx, seed):
    og, sid, eid, iters, predict, fixed, rng_seed, stream, par = draw_case(seed)
    run_variant(ctx, og, sid, eid, iters, predict, fixed, rng_seed, stream, par, nb=26)


@pytest.mark.parametrize("seed,nb", [(s, 6) for s in range(300, 324)] + [(s, 26) for s in range(324, 336)])
def test_random_search_in_ref_mode_equals_the_oracle(ctx, seed, nb):
    """WA_RNG_REF: the reference's own glibc rand() stream and libstdc++ sort order, one ant after another (the oracle's REF mode is what
    tests/test_oracle_golden.py holds against the reference itself).  The reference's default parameters: alpha 1, beta 0.6."""
    og, sid, eid, iters, predict, fixed, rng_seed, stream, par = draw_case(seed)
    par["alpha"] = 1
    iters = min(iters, 12)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    bound = fixed if fixed else int(0.35 * predict / float(og.precision))
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=max(bound, 1), neighbourhood=nb)
    s.srand(rng_seed & 0x7FFFFFFF)
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_REF, **par)
    s.init_pheromone(par["pheromone_0"])
    s.solve(p, sid, eid)
    a = O.Acs(og, nb=nb, pheromone_0=par["pheromone_0"])
    rng = O.srand(rng_seed & 0x7FFFFFFF)
    tr = a.solve(sid, eid, iters, predict, fixed_colony=fixed, mode=O.REF, rng=rng, alpha=par["alpha"], beta=par["beta"], rho=par["rho"],
                 pheromone_0=par["pheromone_0"])
    t = s.trace()
    assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(t["colony"], tr["colony"]) and np.array_equal(t["finite"], tr["finite"])
    assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(bits(t["iterbestL"]), bits(tr["iterbestL"]))
    cost, path, ch = s.result()
    assert bits(cost) == bits(a.best_L)
    if np.isfinite(cost):
        assert np.array_equal(path, a.best_path()[0]) and np.array_equal(ch.astype(np.int32), a.best_path()[1])
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
    st = s.rand_state()   # the libc stream is where the oracle's is
    assert [int(v) for v in st[:31]] == [int(v) for v in rng.r[:31]] and (int(st[34]), int(st[35])) == (int(rng.f), int(rng.b))   # 31 state words, front / back index
    s.close()
    dg.close()
