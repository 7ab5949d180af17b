"""The block that publishes a generation's result (k_evap_rank_mark, block 0 of a search: ACSRank_3D.hpp:263-264 `best = agentK` + the
ranking of :273-275) compares the iteration's best with the global best and then replaces it.  Its four wavefronts must all see the OLD
global best when they decide whether to copy the new path: a wavefront that ran late used to read what thread 0 had just published,
took the improvement for none and left its 64 words of bestpath[] as they were (round 6; found with tools/state_hash.py, once in ~3 000
searches of a saturated batch, never on a lone search).  The knobs build can make wavefronts 1..3 of that block late on purpose
(WA_SWEEP_NT bit 0x400): with it every improving generation of a path longer than 64 nodes would hit the window."""
import os

import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import api, build

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx_knobs():
    assert os.path.exists(build.KNOBS_LIB_PATH), "run __graft_entry__.build() (python -m welding_robot_amd.build --knobs)"
    c = api.Context(0, lib_path=build.KNOBS_LIB_PATH)
    assert b"test knobs" in c.lib.wa_version()
    yield c
    c.close()


def run_against_oracle(ctx, lazy, nb, n=40, ants=48, gens=14):
    og = O.synth_grid(n, seed=77, occ_prob=0.12)
    free = np.nonzero(og.free)[0]
    sid, eid = int(free[0]), int(free[-1])
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ants, lazy=lazy, neighbourhood=nb)
    p = api.default_params(max_iteration=gens, predict=float(3 * n), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=5)
    s.init_pheromone(1.0)
    s.solve(p, sid, eid, streams=[3])
    a = O.Acs(og, nb=nb)
    tr = a.solve(sid, eid, gens, float(3 * n), fixed_colony=ants, mode=O.DEV, seed=5, stream=3)
    t = s.trace()
    cost, path, _ = s.result(0)
    ok = (np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and bits(cost) == bits(a.best_L)
          and np.array_equal(path, a.best_path()[0]) and np.array_equal(bits(s.pheromone()), bits(a.pheromone())))
    improvements = int(np.count_nonzero(np.diff(np.asarray(tr["bestL"], np.float64)) < 0)) + 1
    s.close()
    dg.close()
    return ok, improvements, len(a.best_path()[0])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lazy", [False, True], ids=["dense sweep", "lazy evaporation"])
@pytest.mark.parametrize("nb", [6, 26])
def test_late_wavefronts_of_the_publishing_block_still_copy_their_part_of_the_best_path(ctx_knobs, lazy, nb):
    old = os.environ.get("WA_SWEEP_NT")
    os.environ["WA_SWEEP_NT"] = str(0x400)      # (bits 0-1 = 0: the plain sweep; read when the solver is created)
    try:
        ok, improvements, nodes = run_against_oracle(ctx_knobs, lazy, nb)
    finally:
        if old is None:
            os.environ.pop("WA_SWEEP_NT", None)
        else:
            os.environ["WA_SWEEP_NT"] = old
    assert improvements >= 2 and nodes > 64      # the case does exercise a copy of more than one wavefront's share
    assert ok


@pytest.mark.timeout(300)
def test_the_knob_does_open_the_window():
    """the negative half: the knobs build with the reads where they used to be (-DWA_BEST_READ_LATE, lib/libweldacs_knobs_late_read.so, built by
    __graft_entry__.build()) loses path words under the knob"""
    lib = build.LATE_READ_LIB_PATH
    assert os.path.exists(lib), "run __graft_entry__.build() (python -m welding_robot_amd.build --knobs)"
    c = api.Context(0, lib_path=lib)
    old = os.environ.get("WA_SWEEP_NT")
    os.environ["WA_SWEEP_NT"] = str(0x400)
    try:
        ok, _, _ = run_against_oracle(c, False, 6)
    finally:
        if old is None:
            os.environ.pop("WA_SWEEP_NT", None)
        else:
            os.environ["WA_SWEEP_NT"] = old
        c.close()
    assert not ok
