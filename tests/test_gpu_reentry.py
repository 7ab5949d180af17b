"""Ants that left the best path and stand on it again go back onto the replay track (wa_walk_one / wa_replay_from,
walk_loop_gfx950.hpp VARIANT 2).  The reference has no such thing -- every ant evaluates every step (ACSRank_3D.hpp:
140-197) -- so the only acceptable result is the oracle's, for every ant: path, L, trace, field.  The product switches
the watch on once the best path has been stable for 16 generations; here it is forced on from generation 0, and the
hand-back is also forced on voxels that are NOT on the best path, after very few steps, to hit block boundaries
(64 path words), probe collisions and the arrival at every phase."""
import os

import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import api, build

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx_product():
    c = api.Context(0)
    assert b"test knobs" not in c.lib.wa_version()
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_knobs():
    # the forced-hand-back knobs (WA_REENTRY=2, WA_REENTRY_ANYWHERE, WA_REENTRY_HOLD) are compiled only into this build
    assert os.path.exists(build.KNOBS_LIB_PATH), "run __graft_entry__.build() (python -m welding_robot_amd.build --knobs)"
    c = api.Context(0, lib_path=build.KNOBS_LIB_PATH)
    assert b"test knobs" in c.lib.wa_version()
    yield c
    c.close()


def needs_knob_build(k):
    return "WA_REENTRY_ANYWHERE" in k or "WA_REENTRY_HOLD" in k or k.get("WA_REENTRY") == "2"


KNOBS = {
    "on from generation 0": dict(WA_REENTRY_STABLE="0"),
    "hand back only, anywhere, every 2+ steps": dict(WA_REENTRY_STABLE="0", WA_REENTRY="2", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="2"),
    "hand back anywhere, re-enter when on the path": dict(WA_REENTRY_STABLE="0", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="3"),
    "first hand back after 61 steps": dict(WA_REENTRY_STABLE="0", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="61"),
    # tiny tabu hash (spills to the bitmap after 96 nodes): probe chains longer than the longest hold-off (32) -- the watch
    # must count steps, not evaluation passes, or an ant on the best path never gets its step done
    "tiny hash, long probe chains": dict(WA_REENTRY_STABLE="0", WA_HASH_LOG2="7"),
    "tiny hash, hand back anywhere": dict(WA_REENTRY_STABLE="0", WA_HASH_LOG2="7", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="2"),
    "no touch loads": dict(WA_REENTRY_STABLE="0", WA_WALK_WARM="0", WA_WALK_DIRECT="0", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="5"),
    "no look-ahead (the loop of saturated launches)": dict(WA_REENTRY_STABLE="0", WA_WALK_DIRECT="1", WA_REENTRY_ANYWHERE="1", WA_REENTRY_HOLD="5"),
    "no look-ahead, tiny hash": dict(WA_REENTRY_STABLE="0", WA_WALK_DIRECT="1", WA_HASH_LOG2="7"),
    "product default": dict(),
    "off": dict(WA_REENTRY="0"),
}


@pytest.mark.timeout(300)
@pytest.mark.parametrize("knobs", list(KNOBS), ids=list(KNOBS))
@pytest.mark.parametrize("lazy", [False, True], ids=["dense sweep", "lazy evaporation"])
@pytest.mark.parametrize("n,occ,ants,gens", [(40, 0.12, 48, (3, 14, 45)), (64, 0.0, 64, (30,))])
def test_every_ant_every_path_word_equals_the_oracle(ctx_product, ctx_knobs, knobs, n, occ, ants, gens, lazy):
    ctx = ctx_knobs if needs_knob_build(KNOBS[knobs]) else ctx_product   # everything else runs the shipping library
    og = O.synth_grid(n, seed=77, occ_prob=occ)
    free = np.nonzero(og.free)[0]
    sid, eid = int(free[0]), int(free[-1])
    old = {k: os.environ.get(k) for k in ("WA_REENTRY", "WA_REENTRY_STABLE", "WA_REENTRY_ANYWHERE", "WA_REENTRY_HOLD", "WA_HASH_LOG2", "WA_WALK_WARM", "WA_WALK_DIRECT")}
    os.environ.update(KNOBS[knobs])
    try:
        dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
        for g in gens:
            s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ants, lazy=lazy)   # (the knobs are read when the solver is created)
            p = api.default_params(max_iteration=g, predict=float(3 * n), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=5)
            s.init_pheromone(1.0)
            s.solve(p, sid, eid, streams=[3])
            a = O.Acs(og)
            tr = a.solve(sid, eid, g, float(3 * n), fixed_colony=ants, mode=O.DEV, seed=5, stream=3)
            t = s.trace()
            assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"]))
            L, lens = s.ants()
            olens, oL = a.last_ants()
            assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))
            for i, op in enumerate(a.last_paths()):
                assert np.array_equal(s.ant_path(i), op), (g, i)
            assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
            s.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
