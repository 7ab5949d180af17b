"""Randomised differential sweep: seeded random small problems (grid shape, obstacle density, voxel pitch, alpha, beta,
rho, pheromone_0, colony rule, generations, hash size) solved by the dense solver, the lazily evaporating solver and
the 26-neighbour solver, each against the C oracle -- traces, best path and the full pheromone field bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import api

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def random_case(rs):
    nx, ny, nz = (int(v) for v in rs.randint(1, 14, size=3))
    pitch = float(np.float32(rs.choice([1.0, 0.25, 0.0219, 3.0])))
    free = (rs.uniform(size=nx * ny * nz) >= rs.choice([0.0, 0.1, 0.3])).astype(np.uint8)
    fr = np.flatnonzero(free)
    if len(fr) < 2:
        free[:] = 1
        fr = np.arange(len(free))
    sid, eid = (int(v) for v in rs.choice(fr, size=2, replace=len(fr) < 2))
    og = O.Grid(np.arange(nx, dtype=np.float32) * np.float32(pitch), np.arange(ny, dtype=np.float32) * np.float32(pitch),
                np.arange(nz, dtype=np.float32) * np.float32(pitch), free, pitch, 0)
    par = dict(alpha=int(rs.choice([1, 1, 2, 0, 3])), beta=float(np.float32(rs.choice([0.6, 0.0, 1.5, 0.95]))),
               rho=float(np.float32(rs.choice([0.8, 0.5, 0.99, 1.0]))), pheromone_0=float(np.float32(rs.choice([1.0, 0.3, 7.0]))))
    fixed = int(rs.choice([0, 0, 5, 12, 40]))
    predict = float(np.float32(rs.uniform(2, 30) * pitch))
    if fixed == 0 and int(0.35 * predict / pitch) < 1:
        fixed = 3
    iters = int(rs.choice([1, 3, 10, 25, 60]))
    return og, sid, eid, par, fixed, predict, iters, int(rs.randint(0, 1 << 30)), int(rs.randint(0, 50))


def check(ctx, og, sid, eid, par, fixed, predict, iters, seed, stream, nb, lazy):
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    bound = fixed if fixed else int(0.35 * predict / float(og.precision))
    s = api.AcsSolver(ctx, dg, 1, max(bound, 1), neighbourhood=nb, lazy=lazy)
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_DEV, seed=seed, **par)
    s.init_pheromone(par["pheromone_0"])
    s.solve(p, sid, eid, streams=[stream])
    a = O.Acs(og, pheromone_0=par["pheromone_0"], nb=nb)
    tr = a.solve(sid, eid, iters, predict, fixed_colony=fixed, mode=O.DEV, seed=seed, stream=stream, **par)
    t = s.trace()
    assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(t["finite"], tr["finite"])
    assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(t["colony"], tr["colony"])
    cost, path, ch = s.result()
    assert bits(cost) == bits(a.best_L)
    if np.isfinite(cost):
        assert np.array_equal(path, a.best_path()[0]) and np.array_equal(ch.astype(np.int32), a.best_path()[1])
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
    s.close()
    dg.close()


@pytest.mark.parametrize("chunk", range(16))
def test_random_problems_dense_lazy_and_26(ctx, chunk):
    rs = np.random.RandomState(1000 + chunk)
    for i in range(10):
        og, sid, eid, par, fixed, predict, iters, seed, stream = random_case(rs)
        hl = str(int(rs.choice([6, 8, 11])))
        os.environ["WA_HASH_LOG2"] = hl          # small tables: the bitmap spill path gets its share
        try:
            bound = fixed if fixed else int(0.35 * predict / float(og.precision))
            lazy_ok = bound <= 2048 and int(0.2 * bound) + 1 <= 64
            for nb, lazy in ((6, False), (6, True), (26, False)):
                if lazy and not lazy_ok:
                    continue
                try:
                    check(ctx, og, sid, eid, par, fixed, predict, iters, seed, stream, nb, lazy)
                except AssertionError as e:
                    raise AssertionError("chunk %d case %d nb %d lazy %s dims %dx%dx%d par %s fixed %d predict %g iters %d hash %s: %s" % (
                        chunk, i, nb, lazy, og.nx, og.ny, og.nz, par, fixed, predict, iters, hl, e))
        finally:
            del os.environ["WA_HASH_LOG2"]
