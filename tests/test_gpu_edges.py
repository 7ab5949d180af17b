"""Edge cases of the ACS path on the GPU against the C oracle (DEV mode, bit-exact): degenerate
inputs, parameter ranges the fast paths do not cover, and the slower generic kernels."""
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


def box_grid(nx, ny, nz, occ_prob=0.0, seed=1, p=1.0):
    rs = np.random.RandomState(seed)
    free = (rs.uniform(size=nx * ny * nz) >= occ_prob).astype(np.uint8)
    return O.Grid(np.arange(nx, dtype=np.float32) * p, np.arange(ny, dtype=np.float32) * p, np.arange(nz, dtype=np.float32) * p,
                  free, p, 0)


def run_both(ctx, og, sid, eid, iters, predict, fixed=0, seed=3, stream=0, max_colony=None, path_capacity=0, **par):
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    bound = fixed if fixed else int(0.35 * predict / float(og.precision))
    s = api.AcsSolver(ctx, dg, 1, max_colony or max(bound, 1), path_capacity)
    p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_DEV, seed=seed, **par)
    s.init_pheromone(par.get("pheromone_0", 1.0))
    s.solve(p, sid, eid, streams=[stream])
    a = O.Acs(og, pheromone_0=par.get("pheromone_0", 1.0))
    tr = a.solve(sid, eid, iters, predict, fixed_colony=fixed, mode=O.DEV, seed=seed, stream=stream,
                 alpha=par.get("alpha", 1), beta=par.get("beta", 0.6), rho=par.get("rho", 0.8), pheromone_0=par.get("pheromone_0", 1.0))
    t = s.trace()
    if iters:
        assert np.array_equal(t["steps"], tr["steps"])
        assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(t["finite"], tr["finite"])
        assert np.array_equal(t["colony"], tr["colony"])
    cost, path, ch = s.result()
    assert bits(cost) == bits(a.best_L)
    if np.isfinite(cost):
        assert np.array_equal(path, a.best_path()[0]) and np.array_equal(ch.astype(np.int32), a.best_path()[1])
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
    return s, a, t


def test_start_equals_end_never_arrives(ctx):
    og = box_grid(8, 8, 8)
    s, a, t = run_both(ctx, og, 100, 100, 5, 20.0, fixed=8)
    assert np.isinf(a.best_L) and np.all(t["finite"] == 0)
    assert s.result()[1].size == 0  # cost +inf is a valid result with an empty path


def test_walled_in_start_dies_immediately(ctx):
    og = box_grid(6, 6, 6)
    f = og.free.reshape(6, 6, 6)
    f[2, 2, 3] = f[2, 2, 1] = f[2, 3, 2] = f[2, 1, 2] = f[3, 2, 2] = f[1, 2, 2] = 0  # all six neighbours of (2,2,2)
    sid = (2 * 6 + 2) * 6 + 2
    s, a, t = run_both(ctx, og, sid, 215, 4, 20.0, fixed=6)
    assert np.all(t["steps"] == 0) and np.isinf(a.best_L)


@pytest.mark.parametrize("dims", [(1, 9, 9), (9, 1, 9), (9, 9, 1), (1, 1, 12), (2, 2, 2)])
def test_thin_grids(ctx, dims):
    nx, ny, nz = dims
    og = box_grid(nx, ny, nz, occ_prob=0.1 if nx * ny * nz > 20 else 0.0, seed=nx + 2 * ny)
    og.free[0] = og.free[-1] = 1
    run_both(ctx, og, 0, nx * ny * nz - 1, 12, 30.0, fixed=10, seed=9)


def test_zero_colony_and_zero_iterations(ctx):
    og = box_grid(6, 6, 6)
    run_both(ctx, og, 0, 215, 3, 1.0)            # colony = (int)(0.35*1/1) = 0 ants: nothing ever moves
    run_both(ctx, og, 0, 215, 0, 30.0, fixed=4)  # no generation at all


@pytest.mark.parametrize("par", [dict(alpha=2), dict(alpha=3, beta=0.3), dict(alpha=0), dict(rho=0.5, beta=0.9),
                                 dict(pheromone_0=0.25)])
def test_parameter_ranges(ctx, par):
    og = box_grid(14, 12, 10, occ_prob=0.15, seed=4)
    og.free[0] = og.free[-1] = 1
    run_both(ctx, og, 0, og.n - 1, 25, 40.0, fixed=24, seed=21, **par)


def test_non_unit_precision_and_anisotropic_coords(ctx):
    og = box_grid(10, 11, 12, occ_prob=0.1, seed=6, p=0.0173)
    og.free[0] = og.free[-1] = 1
    run_both(ctx, og, 0, og.n - 1, 30, 0.9, seed=5)  # adaptive colony (Q1) at p = 0.0173


def test_more_than_64_depositing_ranks_uses_chunked_deposit(ctx):
    """lambda - 1 > 64 needs colony >= 326: the deposit runs in 64-rank chunks and the unfused sequence."""
    og = box_grid(12, 12, 12, occ_prob=0.1, seed=8)
    og.free[0] = og.free[-1] = 1
    s, a, t = run_both(ctx, og, 0, og.n - 1, 8, 10.0, fixed=400, seed=2)
    assert t["finite"].max() > 64


def test_colony_above_lds_rank_limit(ctx):
    """> 2048 ants: ranking reads its keys from global memory (k_rank's generic branch)."""
    og = box_grid(10, 10, 10, occ_prob=0.1, seed=9)
    og.free[0] = og.free[-1] = 1
    run_both(ctx, og, 0, og.n - 1, 3, 10.0, fixed=2100, seed=4)


def test_exact_path_capacity_and_reuse_of_a_solver(ctx):
    og = box_grid(5, 1, 1)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, 1.0, 0)
    s = api.AcsSolver(ctx, dg, 1, 4, path_capacity=5)  # the only path 0-1-2-3-4 has exactly 5 nodes
    p = api.default_params(max_iteration=3, predict=10.0, fixed_colony=4, rng_mode=api.RNG_DEV, seed=1)
    s.solve(p, 0, 4)
    cost, path, _ = s.result()
    assert cost == 4.0 and path.tolist() == [0, 1, 2, 3, 4]
    # same solver, new problem after reset(): equals a fresh oracle whose field was reset() the same way
    s.reset_pheromone(1.0)
    s.solve(p, 4, 0)
    a = O.Acs(og)
    a.reset(1.0)
    a.solve(4, 0, 3, 10.0, fixed_colony=4, mode=O.DEV, seed=1, stream=0)
    assert s.result()[1].tolist() == [4, 3, 2, 1, 0] and np.array_equal(bits(s.pheromone()), bits(a.pheromone()))


def test_stepwise_run_equals_single_solve(ctx):
    og = box_grid(12, 12, 12, occ_prob=0.12, seed=3)
    og.free[0] = og.free[-1] = 1
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, 1.0, 0)
    p = api.default_params(max_iteration=30, predict=60.0, fixed_colony=20, rng_mode=api.RNG_DEV, seed=5)
    a = api.AcsSolver(ctx, dg, 1, 20)
    a.solve(p, 0, og.n - 1)
    b = api.AcsSolver(ctx, dg, 1, 20)
    b.begin(p, 0, og.n - 1)
    for chunk in (1, 7, 2, 20):  # odd chunk sizes: the parameter ping-pong must follow the generation parity
        b.run(chunk)
    b.sync()
    assert np.array_equal(bits(a.pheromone()), bits(b.pheromone())) and np.array_equal(a.result()[1], b.result()[1])
    assert np.array_equal(bits(a.trace()["bestL"]), bits(b.trace()["bestL"]))


def test_slot_that_sat_out_an_odd_number_of_generations_is_reused_without_reset(ctx):
    """The sweep is double-buffered and flips the current buffer for the whole solver, but only the ACTIVE slots are
    swept: after an odd number of generations an idle slot's field still lives in the other buffer.  A later search
    that uses it without re-initialising must see that field (here: slot 1's init values), not the other buffer's
    stale contents -- which were all-zero, i.e. 'every edge free and in bounds'."""
    og = box_grid(12, 10, 9, occ_prob=0.12, seed=5)
    og.free[0] = og.free[-1] = 1
    n = 12 * 10 * 9
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=2, max_colony=12)
    s.init_pheromone(1.0)
    p1 = api.default_params(max_iteration=7, predict=40.0, fixed_colony=12, rng_mode=api.RNG_DEV, seed=21)   # odd
    s.solve(p1, [0], [n - 1], streams=[5])
    p2 = api.default_params(max_iteration=6, predict=40.0, fixed_colony=12, rng_mode=api.RNG_DEV, seed=22)
    s.solve(p2, [0, n - 1], [n - 1, 0], streams=[8, 9])                                                      # no reset in between
    a0, a1 = O.Acs(og), O.Acs(og)
    a0.solve(0, n - 1, 7, 40.0, fixed_colony=12, mode=O.DEV, seed=21, stream=5)
    tr0 = a0.solve(0, n - 1, 6, 40.0, fixed_colony=12, mode=O.DEV, seed=22, stream=8)   # carries slot 0's field over
    tr1 = a1.solve(n - 1, 0, 6, 40.0, fixed_colony=12, mode=O.DEV, seed=22, stream=9)   # slot 1: still the init field
    for q, (a, tr) in enumerate(((a0, tr0), (a1, tr1))):
        t = s.trace(q)
        assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])), q
        assert np.array_equal(bits(s.pheromone(q)), bits(a.pheromone())), q
    # ... and an idle slot can still be read back while the others moved on (an odd number of generations again)
    s.solve(p1, [0], [n - 1], streams=[5])
    assert np.array_equal(bits(s.pheromone(1)), bits(a1.pheromone()))
    # ... and the stand-alone sweep (wa_acs_evaporate flips every slot) must carry the idle slot's LIVE field across, not the
    # stale copy in the buffer it was last swept out of: slot 1 unchanged, slot 0 multiplied by rho once per repeat
    f0 = s.pheromone(0)
    s.evaporate(0, 0.5, 1)
    assert np.array_equal(bits(s.pheromone(1)), bits(a1.pheromone()))
    assert np.array_equal(bits(s.pheromone(0)), bits(f0 * np.float32(0.5)))
    s.evaporate(1, 0.25, 3)
    assert np.array_equal(bits(s.pheromone(1)), bits(((a1.pheromone() * np.float32(0.25)) * np.float32(0.25)) * np.float32(0.25)))
    assert np.array_equal(bits(s.pheromone(0)), bits(f0 * np.float32(0.5)))


def test_every_ant_of_a_generation_matches_the_oracle(ctx):
    """agents[] of the last generation (ACSRank_3D.hpp:251-261): L and node count per ant, not only the best."""
    og = box_grid(20, 17, 15, occ_prob=0.1, seed=3)
    og.free[0] = og.free[-1] = 1
    n = 20 * 17 * 15
    for gens in (1, 4):
        s, a, t = run_both(ctx, og, 0, n - 1, gens, 90.0, fixed=31, seed=13, stream=2)
        L, lens = s.ants()
        olens, oL = a.last_ants()
        assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))


def test_hand_scheduled_walk_loop_equals_the_compiler_scheduled_one(ctx):
    """WA_WALK_ASM=0 keeps the C++ loop; WA_WALK_WARM=0 / 1 forces the hand-scheduled loop without / with its touch loads,
    WA_WALK_DIRECT=1 the loop without any look-ahead (the product picks by launch size): all must give the same ants, the
    same field, the same trace."""
    og = box_grid(24, 24, 24, occ_prob=0.1, seed=8)
    og.free[0] = og.free[-1] = 1
    n = 24 ** 3
    res = []
    for knobs in (dict(WA_WALK_ASM="1"), dict(WA_WALK_ASM="0"), dict(WA_WALK_WARM="0", WA_WALK_DIRECT="0"), dict(WA_WALK_WARM="1"), dict(WA_WALK_DIRECT="1")):
        os.environ.update(knobs)
        try:
            s, a, t = run_both(ctx, og, 0, n - 1, 25, 200.0, fixed=64, seed=77)
            res.append((s.ants(), s.pheromone(), t))
        finally:
            for k in knobs:
                del os.environ[k]
    (x, px, tx) = res[0]
    for (y, py, ty) in res[1:]:
        assert np.array_equal(x[1], y[1]) and np.array_equal(bits(x[0]), bits(y[0])) and np.array_equal(bits(px), bits(py))
        assert np.array_equal(tx["steps"], ty["steps"])


def test_searches_with_the_same_end_point_share_one_heuristic_field(ctx):
    """wa_acs_begin computes one (1 + beta*cos) field (ACSRank_3D.hpp:151-154) per distinct END point of a batch and keeps a
    slot's field across batches while end point and beta stay the same.  Every search must still equal its own oracle run:
    duplicates inside a batch, a second batch that reuses and re-assigns fields, a changed beta, a changed end point."""
    og = box_grid(14, 12, 11, occ_prob=0.1, seed=9)
    og.free[0] = og.free[-1] = og.free[5] = og.free[100] = 1
    n = 14 * 12 * 11
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=4, max_colony=10)
    batches = [
        (0.6, [0, 5, 100, 5], [n - 1, n - 1, 0, 0]),        # slots 0,1 share; 2,3 share
        (0.6, [100, 0, 5, 0], [n - 1, 0 + 5, 0, n - 1]),    # slot 0 keeps its field, 1 needs a new one, 2 keeps, 3 reads slot 0's
        (0.9, [0, 5], [n - 1, n - 1]),                      # another beta: nothing may be kept
        (0.9, [5, 0], [0, n - 1]),                          # slot 0's end point changes, slot 1 keeps
    ]
    for b, (beta, starts, ends) in enumerate(batches):
        p = api.default_params(max_iteration=9, predict=30.0, fixed_colony=10, rng_mode=api.RNG_DEV, seed=40 + b, beta=beta)
        s.init_pheromone(1.0)
        s.solve(p, starts, ends, streams=list(range(10 * b, 10 * b + len(starts))))
        for q, (sid, eid) in enumerate(zip(starts, ends)):
            a = O.Acs(og)
            tr = a.solve(sid, eid, 9, 30.0, fixed_colony=10, mode=O.DEV, seed=40 + b, stream=10 * b + q, beta=beta)
            t = s.trace(q)
            assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])), (b, q)
            assert np.array_equal(bits(s.pheromone(q)), bits(a.pheromone())), (b, q)
    s.close()


def test_heuristic_field_pool_grows_and_recycles(ctx):
    """The pool starts with four fields: a batch with six distinct end points grows it (keeping what it holds), later batches
    recycle the fields that the current batch does not use, oldest first.  Every search against its own oracle run."""
    og = box_grid(13, 11, 10, occ_prob=0.08, seed=4)
    n = 13 * 11 * 10
    free = np.nonzero(og.free)[0]
    pts = [int(free[i]) for i in (0, 7, 50, 200, 400, 700, 900, len(free) - 1)]
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=6, max_colony=8)
    batches = [
        ([0, 1, 2], [7, 7, 6]),                    # two end points
        ([0, 1, 2, 3, 4, 6], [7, 6, 5, 4, 3, 2]),  # six: the pool grows from four to six fields, 7 and 6 are kept
        ([1, 2, 3, 4], [0, 0, 1, 7]),              # new end points recycle the oldest fields; 7 is still there
        ([5, 4, 3, 2, 1, 0], [6, 5, 4, 3, 2, 1]),
    ]
    for b, (st, en) in enumerate(batches):
        p = api.default_params(max_iteration=7, predict=25.0, fixed_colony=8, rng_mode=api.RNG_DEV, seed=90 + b)
        s.init_pheromone(1.0)
        s.solve(p, [pts[i] for i in st], [pts[i] for i in en], streams=list(range(len(st))))
        for q in range(len(st)):
            a = O.Acs(og)
            tr = a.solve(pts[st[q]], pts[en[q]], 7, 25.0, fixed_colony=8, mode=O.DEV, seed=90 + b, stream=q)
            t = s.trace(q)
            assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])), (b, q)
            assert np.array_equal(bits(s.pheromone(q)), bits(a.pheromone())), (b, q)
    s.close()


def test_byte_rank_masks_equal_u64_rank_masks(ctx):
    """A solver whose colonies never exceed 35 ants (at most 8 depositing ranks, ACSRank_3D.hpp:200) keeps its deposit rank
    masks in one byte per edge; WA_MASK_U64=1 keeps the u64 masks.  Dense and lazy, 8 ranks exactly (35 ants), shared words
    (edges of one voxel sit in the same 32-bit word the marks OR into)."""
    og = box_grid(20, 16, 12, occ_prob=0.08, seed=12)
    og.free[0] = og.free[-1] = 1
    n = 20 * 16 * 12
    res = {}
    for lazy in (False, True):
        for wide in ("0", "1"):
            os.environ["WA_MASK_U64"] = wide
            try:
                dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
                s = api.AcsSolver(ctx, dg, n_slots=2, max_colony=35, lazy=lazy)
                p = api.default_params(max_iteration=30, predict=100.0, fixed_colony=35, rng_mode=api.RNG_DEV, seed=3)
                s.init_pheromone(1.0)
                s.solve(p, [0, n - 1], [n - 1, 0], streams=[4, 9])
                res[(lazy, wide)] = [(s.trace(q), s.pheromone(q)) for q in range(2)]
                s.close()
            finally:
                del os.environ["WA_MASK_U64"]
    a = O.Acs(og)
    tr = a.solve(0, n - 1, 30, 100.0, fixed_colony=35, mode=O.DEV, seed=3, stream=4)
    for key, slots in res.items():
        t, ph = slots[0]
        assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])), key
        assert np.array_equal(bits(ph), bits(a.pheromone())), key
        for q in range(2):
            assert np.array_equal(bits(slots[q][1]), bits(res[(False, "1")][q][1])), (key, q)


def test_profile_can_stamp_the_sweep_launch_of_every_generation(ctx):
    """wa_acs_profile(enable = 3): every n-th generation has all its launches stamped, the launch that carries the evaporation
    sweep is stamped in EVERY generation (per-dispatch events, no extra stream operation)."""
    og = box_grid(16, 16, 16, occ_prob=0.1, seed=2)
    og.free[0] = og.free[-1] = 1
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, 1.0, 0)
    s = api.AcsSolver(ctx, dg, 1, 16)
    p = api.default_params(max_iteration=20, predict=60.0, fixed_colony=16, rng_mode=api.RNG_DEV, seed=4)
    s.begin(p, 0, 16 ** 3 - 1)
    s.profile(True, 10, sweep_every_generation=True)
    s.run(20)
    pr = s.profile_read()
    assert pr["evaporate"]["launches"] == 20 and pr["walk"]["launches"] == 2 and pr["deposit"]["launches"] == 2
    assert pr["evaporate"]["ms"] > 0
    s.profile(True, 5)
    s.run(0)
    s.close()


_strag_oracle = {}


def _strag_want(og, n, nb, stream, iters=14, seed=77):
    key = (nb, stream, iters, seed)
    if key not in _strag_oracle:
        a = O.Acs(og, nb=nb)
        tr = a.solve(0, n - 1, iters, 132.0, fixed_colony=96, mode=O.DEV, seed=seed, stream=stream)
        olens, oL = a.last_ants()
        paths = a.last_paths()
        _strag_oracle[key] = (tr["steps"].copy(), tr["finite"].copy(), bits(tr["bestL"]).copy(), bits(oL).copy(), olens.copy(), bits(a.pheromone()).copy(), paths)
    return _strag_oracle[key]


@pytest.mark.parametrize("P", [1, 2, 8])
@pytest.mark.parametrize("nb", [6, 26])
def test_stragglers_are_handed_over_and_nothing_changes(ctx, monkeypatch, nb, P):
    """Dense searches hand ants that can no longer matter to resume blocks of the next walk launch (include/weldacs.h,
    wa_acs_debug_counters), every search of a batch its own (lists and pools per slot: P = 1, 2, 8 searches in one solver).  With the
    mechanism on, off, run generation by generation with a read in between (the last generation of a call hands over too and a drain launch
    finishes its stragglers) and the same with WA_STRAGGLER_DRAIN=0 (where it then never engages): the same trace -- steps and finite ants
    included --, the same ants in the last generation, the same field, all equal to the oracle.  26 neighbours: the arrivals are compared
    by their L (step lengths differ per move type), a resumed ant continues its in-order fp32 sum."""
    og = box_grid(48, 40, 44, occ_prob=0.1, seed=21)
    og.free[0] = og.free[-1] = 1
    n = 48 * 40 * 44
    out = np.zeros(16, np.uint64)
    streams = [4 + q for q in range(P)]
    for mode in ("on", "off", "stepwise", "stepwise_nodrain"):
        monkeypatch.setenv("WA_STRAGGLERS", "0" if mode == "off" else "1")
        monkeypatch.setenv("WA_STRAGGLER_DRAIN", "0" if mode == "stepwise_nodrain" else "1")
        dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
        s = api.AcsSolver(ctx, dg, P, 96, neighbourhood=nb)
        p = api.default_params(max_iteration=14, predict=132.0, fixed_colony=96, rng_mode=api.RNG_DEV, seed=77)
        s.init_pheromone(1.0)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        if mode.startswith("stepwise"):
            s.begin(p, [0] * P, [n - 1] * P, streams=streams)
            for g in range(14):
                s.run(1)
                if g % 3 == 0:      # a read between the calls: complete whenever it happens
                    L, lens = s.ants(P - 1)
                    assert np.all(lens >= 1) and (np.isinf(L) | (lens > 1)).all()
            s.sync()
        else:
            s.solve(p, [0] * P, [n - 1] * P, streams=streams)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
        handed, resumed = int(out[9]), int(out[7])
        per_slot = [s.straggler_counters(q) for q in range(P)]
        assert sum(h for h, _ in per_slot) == handed and all(h == r for h, r in per_slot)
        if mode in ("on", "stepwise"):
            assert handed > 0 and resumed == handed       # it engaged, and every straggler was finished
            assert all(h > 0 for h, _ in per_slot)        # ... in every search of the batch
        else:
            assert handed == 0 and resumed == 0
        for q in range(P):
            t = s.trace(q)
            L, lens = s.ants(q)
            got = (t["steps"], t["finite"], bits(t["bestL"]), bits(L), lens, bits(s.pheromone(q)))
            want = _strag_want(og, n, nb, streams[q])
            for g_, w in zip(got, want[:6]):
                assert np.array_equal(g_, w), (mode, q)
            if q == P - 1:   # every ant's whole path: the ants the LAST generation handed over were finished by the drain launch and put back
                for i, op in enumerate(want[6]):
                    assert np.array_equal(s.ant_path(i, q), op), (mode, q, i)
        s.close()
        dg.close()


@pytest.mark.parametrize("nb", [6, 26])
@pytest.mark.parametrize("chunks", [(2,), (3,), (2, 2, 1, 3), (5, 1, 4)])
def test_straggler_hand_over_across_run_boundaries(ctx, chunks, nb):
    """wa_acs_run in pieces: the last generation of every call hands nothing over, the others may; the search is the oracle's whatever the split."""
    og = box_grid(48, 40, 44, occ_prob=0.1, seed=21)
    og.free[0] = og.free[-1] = 1
    n, total = 48 * 40 * 44, sum(chunks)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, 1, 96, neighbourhood=nb)
    p = api.default_params(max_iteration=total, predict=132.0, fixed_colony=96, rng_mode=api.RNG_DEV, seed=78)
    s.init_pheromone(1.0)
    s.begin(p, 0, n - 1, streams=[2])
    for c in chunks:
        s.run(c)
        s.sync()
        out = np.zeros(16, np.uint64)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
        assert int(out[9]) == int(out[7])                 # nothing is left over between calls
    a = O.Acs(og, nb=nb)
    tr = a.solve(0, n - 1, total, 132.0, fixed_colony=96, mode=O.DEV, seed=78, stream=2)
    t = s.trace()
    assert np.array_equal(t["steps"][:total], tr["steps"]) and np.array_equal(t["finite"][:total], tr["finite"])
    assert np.array_equal(bits(t["bestL"][:total]), bits(tr["bestL"]))
    L, lens = s.ants()
    olens, oL = a.last_ants()
    assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
    s.close()
    dg.close()


def test_results_read_between_runs_and_between_problems_are_never_stale(ctx):
    """wa_acs_result / wa_acs_result_batch serve a host copy fetched once behind a run (host_acs.inc: acs_fetch_results): it has to be
    dropped by the next wa_acs_run and the next wa_acs_begin.  Read after every 3 generations of a 24-generation search in 4 slots, per
    slot and as a batch, against the oracle's state at that generation; then another problem on the same solver."""
    og = box_grid(14, 12, 10, occ_prob=0.12, seed=5)
    free_ids = np.flatnonzero(og.free)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, 4, 24)
    starts = [int(free_ids[3]), int(free_ids[11]), int(free_ids[40]), int(free_ids[7])]
    ends = [int(free_ids[-5]), int(free_ids[-40]), int(free_ids[-9]), int(free_ids[-77])]
    for round_ in range(2):
        if round_:
            starts, ends = ends[::-1], starts[::-1]
        p = api.default_params(max_iteration=24, predict=40.0, fixed_colony=24, rng_mode=api.RNG_DEV, seed=11 + round_)
        s.init_pheromone(1.0)
        s.begin(p, starts, ends, streams=[0, 1, 2, 3])
        changed = 0
        last = [None] * 4
        for g in range(0, 24, 3):
            s.run(3)
            costs, paths = s.results()
            for q in range(4):
                a = O.Acs(og)
                a.solve(starts[q], ends[q], g + 3, 40.0, fixed_colony=24, mode=O.DEV, seed=11 + round_, stream=q)
                cost, path, ch = s.result(q)
                assert bits(cost) == bits(a.best_L) == bits(costs[q])
                if np.isfinite(cost):
                    assert np.array_equal(path, a.best_path()[0]) and np.array_equal(ch.astype(np.int32), a.best_path()[1])
                    assert np.array_equal(paths[q], path)
                if last[q] is not None and bits(cost) != bits(last[q]):
                    changed += 1
                last[q] = cost
        assert changed > 0      # the best cost did move between reads: a stale copy would have been caught
