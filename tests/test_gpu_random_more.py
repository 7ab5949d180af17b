"""More randomised differential tests against the C oracle (bit-exact), beside tests/test_gpu_random.py: one seeded case per test id --
searches on random box grids incl. every ant of the last generation, REF mode with the libc stream position, batches in the slots of one
solver, medium-size grids (path blocks, replay, rejoin watch), the other code paths (knobs); voxelisation of random meshes; seam
ordering of random distance matrices; B-splines.  Every case is a fixed seed, so a failure reproduces."""
import os

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


@pytest.mark.parametrize("knobs", [{"WA_HASH_LOG2": "6"}, {"WA_HASH_LOG2": "8"}, {"WA_WALK_ASM": "0"}, {"WA_HASH_LOG2": "7", "WA_WALK_ASM": "0"}, {}])
def test_ref_mode_on_the_hand_scheduled_loop_down_its_exits(ctx, knobs, monkeypatch):
    """REF mode runs the hand-scheduled loop with the libc stream generated 64 draws at a time (round 4).  Medium corner-to-corner searches
    (walks of several 64-step blocks: hand-backs at block boundaries, dead ends handed to the generic loop undecided) with tabu tables so small
    that every walk outgrows the fast loop and finishes in the spilled one, with the compiler-scheduled loop instead, and as built: all
    equal to the oracle's REF run down to the libc stream position."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for seed in (11, 12, 13):
        rs = np.random.RandomState(seed)
        nx, ny, nz = (int(rs.randint(24, 44)) for _ in range(3))
        og = box_grid(nx, ny, nz, occ_prob=float(rs.choice([0.05, 0.15, 0.25])), seed=seed)
        og.free[0] = og.free[-1] = 1
        n = nx * ny * nz
        ants, iters, predict = int(rs.randint(8, 40)), 6, float(nx + ny + nz)
        dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ants)
        s.srand(1000 + seed)
        p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=ants, rng_mode=api.RNG_REF)
        s.init_pheromone(1.0)
        s.solve(p, 0, n - 1)
        a = O.Acs(og)
        rng = O.srand(1000 + seed)
        tr = a.solve(0, n - 1, iters, predict, fixed_colony=ants, mode=O.REF, rng=rng)
        t = s.trace()
        assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(t["finite"], tr["finite"]), (knobs, seed)
        assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(bits(t["iterbestL"]), bits(tr["iterbestL"]))
        L, lens = s.ants()
        olens, oL = a.last_ants()
        assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))
        for i, op in enumerate(a.last_paths()):
            assert np.array_equal(s.ant_path(i), op), (knobs, seed, i)
        assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
        st = s.rand_state()
        assert [int(v) for v in st[:31]] == [int(v) for v in rng.r[:31]] and (int(st[34]), int(st[35])) == (int(rng.f), int(rng.b))
        assert int(tr["steps"].max()) > 64 * ants // 4          # the walks really span several blocks
        s.close()
        dg.close()


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_ref_mode_speculates_converged_generations_and_nothing_changes(ctx, seed, monkeypatch):
    """Once a REF colony has converged every ant re-walks the best path, so the draws each ant is dealt are known in advance: the stream of
    the generation is generated ahead, the ants check in parallel that they follow the whole path and the sequential walk starts at the
    first one that does not (include/weldacs.h, wa_acs_debug_counters).  Small open grids converge within a few generations: with the
    speculation on and off the trace, every ant's path, the field and the libc stream position are equal -- and equal to the oracle's REF run
    -- and the speculation confirmed ants."""
    rs = np.random.RandomState(seed)
    nx, ny, nz = (int(rs.randint(10, 22)) for _ in range(3))
    og = box_grid(nx, ny, nz, occ_prob=float(rs.choice([0.0, 0.05])), seed=seed)
    og.free[0] = og.free[-1] = 1
    n = nx * ny * nz
    ants, iters, predict = int(rs.randint(12, 48)), 60, float(nx + ny + nz)
    out = np.zeros(16, np.uint64)
    res = {}
    for spec in ("1", "0"):
        monkeypatch.setenv("WA_REF_SPEC", spec)
        dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
        s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=ants)
        s.srand(77 + seed)
        p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=ants, rng_mode=api.RNG_REF)
        s.init_pheromone(1.0)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        s.solve(p, 0, n - 1)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
        confirmed = int(out[6])
        assert (confirmed > ants) if spec == "1" else (confirmed == 0)       # it engaged (whole generations' worth of ants), or it is off
        t = s.trace()
        L, lens = s.ants()
        st = s.rand_state()
        res[spec] = (t["steps"].copy(), t["finite"].copy(), bits(t["bestL"]).copy(), bits(L).copy(), lens.copy(), [s.ant_path(i).copy() for i in range(ants)],
                     bits(s.pheromone()).copy(), [int(v) for v in st[:31]] + [int(st[34]), int(st[35])])
        s.close()
        dg.close()
    a = O.Acs(og)
    rng = O.srand(77 + seed)
    tr = a.solve(0, n - 1, iters, predict, fixed_colony=ants, mode=O.REF, rng=rng)
    olens, oL = a.last_ants()
    want = (tr["steps"], tr["finite"], bits(tr["bestL"]), bits(oL), olens, a.last_paths(), bits(a.pheromone()), [int(v) for v in rng.r[:31]] + [int(rng.f), int(rng.b)])
    for spec in res:
        for k, (g_, w) in enumerate(zip(res[spec], want)):
            if k == 5:
                assert all(np.array_equal(x, y) for x, y in zip(g_, w)), (spec, "ant paths")
            else:
                assert np.array_equal(g_, w), (spec, k)


# ------------------------------------------------------------------ voxelisation (a3 / N1)
@pytest.mark.parametrize("seed", range(400, 424))
def test_random_mesh_voxelisation_equals_the_oracle(ctx, seed):
    rs = np.random.RandomState(seed)
    nt = int(rs.randint(1, 120))
    v = rs.uniform(-1, 1, (nt, 3, 3)).astype(np.float32) * np.float32(rs.choice([1.0, 0.05, 30.0]))
    k = rs.rand(nt)
    v[k < 0.3] = v[k < 0.3][:, :1] + rs.uniform(-0.06, 0.06, (int((k < 0.3).sum()), 3, 3)).astype(np.float32)   # small triangles
    ax = int(rs.randint(0, 3))
    flat = (k >= 0.3) & (k < 0.45)
    v[flat, :, ax] = v[flat, :1, ax]                                                                              # axis-aligned ones
    if nt > 4:
        v[1] = v[1, 0]            # a point
        v[2, 2] = v[2, 1]         # a segment
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    with np.errstate(invalid="ignore", divide="ignore"):
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
    tris = np.zeros((nt, 12), np.float32)
    tris[:, :3] = n
    tris[:, 3:] = v.reshape(nt, 9)
    ext = float((v.max(axis=(0, 1)) - v.min(axis=(0, 1))).max())
    p = max(ext, 1e-3) / float(rs.randint(3, 40))
    wall = int(rs.randint(0, 6))
    og = O.grid_from_mesh(tris, p, wall)
    a = api.Grid.from_mesh(ctx, tris, p, wall)
    os.environ["WA_VOXELIZE_DENSE"] = "1"
    try:
        b = api.Grid.from_mesh(ctx, tris, p, wall)
    finally:
        del os.environ["WA_VOXELIZE_DENSE"]
    assert (a.nx, a.ny, a.nz) == (og.nx, og.ny, og.nz) == (b.nx, b.ny, b.nz)
    cx, cy, cz = a.coords()
    assert np.array_equal(bits(cx), bits(og.cx)) and np.array_equal(bits(cy), bits(og.cy)) and np.array_equal(bits(cz), bits(og.cz))
    assert np.array_equal(a.occupancy(), og.free) and np.array_equal(b.occupancy(), og.free)
    a.close()
    b.close()


# ------------------------------------------------------------------ seam ordering (a13-a15)
@pytest.mark.parametrize("seed", range(500, 524))
def test_random_seam_ordering_equals_the_oracle(ctx, seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(2, 90))
    P = rs.uniform(0, 1, (n, 3)) * float(rs.choice([1.0, 100.0, 0.01]))
    if n > 6 and rs.rand() < 0.5:
        P[int(rs.randint(0, n))] = P[int(rs.randint(0, n))]            # two cities in one place: zero distance, ties
    d = np.abs(P[:, None, :] - P[None, :, :]).sum(-1) if rs.rand() < 0.5 else np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
    cap = int(rs.choice([0, 0, 5, 17]))
    sd, stream = int(rs.randint(1, 1 << 30)), int(rs.randint(0, 16))
    o = O.gtsp_solve(d, mode=O.DEV, seed=sd, stream=stream, max_iterations=cap, want_pher=True)
    for wave in ("1", "0"):
        os.environ["WA_GTSP_WAVE"] = wave
        try:
            t = api.gtsp_solve(ctx, d, mode=api.RNG_DEV, seed=sd, stream=stream, max_iterations=cap, want_pher=True)
        finally:
            del os.environ["WA_GTSP_WAVE"]
        assert t["iters"][0] == o["iters"] and t["L"][0] == o["L"], (n, wave)
        assert np.array_equal(t["edges"][0], o["edges"]), (n, wave)
        assert np.array_equal(t["pher"][0].view(np.uint64), o["pher"].view(np.uint64)), (n, wave)
    rng = O.srand(sd & 0x7FFFFFFF)                                      # the reference's own stream
    st = np.array(list(rng.r) + [rng.f, rng.b], np.int32)
    o = O.gtsp_solve(d, mode=O.REF, rng=rng, max_iterations=cap or 40)
    t = api.gtsp_solve(ctx, d, mode=api.RNG_REF, rand_state=st, max_iterations=cap or 40)
    assert t["iters"][0] == o["iters"] and t["L"][0] == o["L"] and np.array_equal(t["edges"][0], o["edges"])
    assert [int(v) for v in t["rand_state"][:31]] == list(rng.r)[:31]


# ------------------------------------------------------------------ B-spline smoothing (N3)
def nbits(a):
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).ravel().copy()
    b[np.isnan(a).ravel()] = 0x7fc00000        # NaNs compare equal whatever their sign / payload
    return b


@pytest.mark.parametrize("seed", range(600, 632))
def test_random_bspline_equals_the_oracle(ctx, seed):
    rs = np.random.RandomState(seed)
    dim, deg = int(rs.randint(1, 17)), int(rs.randint(0, 8))
    ci, cf = int(rs.randint(0, deg + 1)), int(rs.randint(0, deg + 1))
    n = int(rs.randint(max(2, deg), 3000))
    tf = float(rs.choice([1.0, 150.0, 6000.0, 0.37]))
    mid = (np.cumsum(rs.uniform(-0.01, 0.01, size=(n, dim)), axis=0) * float(rs.choice([1.0, 1000.0]))).astype(np.float32)
    init = rs.uniform(-1, 1, size=(ci + 1, dim)).astype(np.float32)
    fin = rs.uniform(-1, 1, size=(cf + 1, dim)).astype(np.float32)
    ob = O.Bspline(dim, deg, ci, cf, n)
    ob.set_param(init, fin, mid, tf)
    b = api.Bspline(ctx, dim, deg, ci, cf, n)
    b.set_param(init, fin, mid, tf)
    knots, cps = b.arrays()
    assert np.array_equal(nbits(knots), nbits(ob.knots)) and np.array_equal(nbits(cps), nbits(ob.cps))
    count = int(rs.randint(1, 5000))
    t0, dt = np.float32(-0.02 * tf), np.float32(1.05 * tf / count)
    for der in sorted({0, min(1, deg), deg, deg + 1}):
        got, ok = b.sample(t0, dt, count, der)
        want, wok = ob.sample(t0, dt, count, der, prefill=0.0)
        assert np.array_equal(ok, wok) and np.array_equal(nbits(got), nbits(want)), der
    us = np.concatenate([rs.uniform(-0.5 * tf, 1.5 * tf, size=500), [0.0, tf, tf / 2]]).astype(np.float32)   # incl. both ends exactly
    got, ok = b.eval(us)
    want, wok = ob.eval(us, prefill=0.0)
    assert np.array_equal(ok, wok) and np.array_equal(nbits(got), nbits(want))
    b.close()


# ------------------------------------------------------------------ batches of searches in one solver (a11 / C4 / C5)
@pytest.mark.parametrize("seed", range(700, 716))
def test_random_batch_of_searches_equals_single_oracle_runs(ctx, seed):
    """Several searches advance together in the slots of one solver (different end points, shared heuristic fields where they coincide,
    their own DEV streams), twice in a row on the same solver; each must equal the oracle run on its own."""
    rs = np.random.RandomState(seed)
    nx, ny, nz = (int(rs.randint(4, 21)) for _ in range(3))
    og = box_grid(nx, ny, nz, occ_prob=float(rs.choice([0.0, 0.15, 0.3])), seed=seed, p=float(rs.choice([1.0, 0.25])))
    free = np.flatnonzero(og.free)
    slots = int(rs.randint(2, 10))
    lazy = bool(rs.rand() < 0.5)
    ants = int(rs.randint(2, 40))
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    sb = api.AcsSolver(ctx, dg, n_slots=slots, max_colony=ants, lazy=lazy)
    for rnd in range(2):
        k = slots if rnd == 0 else int(rs.randint(1, slots + 1))         # the second batch may leave slots idle
        ends = rs.choice(free, min(3, len(free)), replace=False)             # few distinct end points: heuristic fields are shared
        starts = [int(v) for v in rs.choice(free, k)]
        endl = [int(v) for v in rs.choice(ends, k)]
        streams = [int(v) for v in rs.randint(0, 1000, k)]
        iters, sd = int(rs.randint(1, 30)), int(rs.randint(1, 1 << 30))
        p = api.default_params(max_iteration=iters, predict=float((nx + ny + nz) * og.precision), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=sd)
        sb.init_pheromone(1.0)   # (initFromGridMap; reset() also fills the out-of-bounds edges, which the pair-flow goldens cover)
        sb.solve(p, starts, endl, streams=streams)
        for q in range(k):
            a = O.Acs(og)
            a.solve(starts[q], endl[q], iters, float((nx + ny + nz) * og.precision), fixed_colony=ants, mode=O.DEV, seed=sd, stream=streams[q])
            cost, path, _ = sb.result(q)
            assert bits(cost) == bits(a.best_L), (rnd, q)
            if np.isfinite(cost):
                assert np.array_equal(path, a.best_path()[0])
            assert np.array_equal(bits(sb.pheromone(q)), bits(a.pheromone())), (rnd, q)
    sb.close()
    dg.close()


@pytest.mark.parametrize("seed", range(800, 816))
def test_random_medium_search_equals_the_oracle(ctx, seed):
    """Corner-to-corner searches on 30..64-voxel grids: walks of a few hundred steps (several 64-word path blocks, table collisions),
    enough generations for the best-path replay and -- from generation 16 on -- the rejoin watch; dense and lazy."""
    rs = np.random.RandomState(seed)
    nx, ny, nz = (int(rs.randint(30, 65)) for _ in range(3))
    og = box_grid(nx, ny, nz, occ_prob=float(rs.choice([0.05, 0.1, 0.2])), seed=seed, p=1.0)
    og.free[0] = og.free[-1] = 1
    par = dict(alpha=1, beta=float(rs.choice([0.6, 1.0])), rho=float(rs.choice([0.8, 0.9])), pheromone_0=1.0)
    run_variant(ctx, og, 0, nx * ny * nz - 1, int(rs.randint(20, 46)), float(nx + ny + nz), int(rs.randint(16, 65)), int(rs.randint(1, 1 << 30)),
                int(rs.randint(0, 8)), par, lazy=bool(seed & 1))


# ------------------------------------------------------------------ the same random searches down the other code paths
KNOBS = [dict(WA_WALK_ASM="0"),                         # the compiler-scheduled loop instead of the hand-scheduled one
         dict(WA_WALK_WARM="0", WA_WALK_DIRECT="0"),    # look-ahead without touch loads
         dict(WA_WALK_DIRECT="1"),                      # no look-ahead at all (the saturated-launch variant of the loop)
         dict(WA_HASH_LOG2="6"),                        # a 64-entry tabu table: probe chains, spill to the global bitmap
         dict(WA_HASH_LOG2="8", WA_WALK_WARM="0", WA_WALK_DIRECT="0"),
         dict(WA_HASH_LOG2="8", WA_WALK_DIRECT="1"),
         dict(WA_WALK_DIRECT="1", WA_REENTRY_STABLE="1"),
         dict(WA_REENTRY="0"),                          # no rejoin watch
         dict(WA_REENTRY_STABLE="1"),                   # ... armed after one stable generation
         dict(WA_EVAP_BLOCKS="7"), dict(WA_LAZY_BLOCKS="3")]   # odd sweep / background-pass grids


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_random_searches_down_the_other_code_paths(ctx, knobs, monkeypatch):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)   # read by wa_acs_create
    for seed in list(range(0, 96, 8)) + [801, 806, 811]:
        if seed < 800:
            og, sid, eid, iters, predict, fixed, rng_seed, stream, par = draw_case(seed)
            run_variant(ctx, og, sid, eid, iters, predict, fixed, rng_seed, stream, par, lazy=bool(seed & 8))
        else:
            rs = np.random.RandomState(seed)
            nx, ny, nz = (int(rs.randint(30, 65)) for _ in range(3))
            og = box_grid(nx, ny, nz, occ_prob=0.1, seed=seed, p=1.0)
            og.free[0] = og.free[-1] = 1
            run_variant(ctx, og, 0, nx * ny * nz - 1, 24, float(nx + ny + nz), 32, seed, 1, dict(alpha=1, beta=0.6, rho=0.8, pheromone_0=1.0),
                        lazy=bool(seed & 1))


# ------------------------------------------------------------------ stragglers on / off on random hand-over-heavy searches
@pytest.mark.parametrize("nb", [6, 26])
@pytest.mark.parametrize("trial", range(16))
def test_random_searches_with_and_without_the_straggler_hand_over(ctx, trial, monkeypatch, nb):
    """Which ants are handed over depends on timing; nothing observable may: trace (steps and finite ants included), the last
    generation's ants and the whole field with the mechanism on equal those with it off, and every straggler is finished."""
    from welding_robot_amd import synth
    rs = np.random.RandomState(trial)
    n = int(rs.choice([48, 64, 96]))
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=100 + trial, occ_prob=float(rs.choice([0.05, 0.1, 0.2])))
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    ants, gens = int(rs.choice([64, 128, 256])), int(rs.randint(12, 40))
    res, out = {}, np.zeros(16, np.uint64)
    for mode in ("1", "0"):
        monkeypatch.setenv("WA_STRAGGLERS", mode)
        s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=ants, neighbourhood=nb)
        p = api.default_params(max_iteration=gens, predict=3.0 * n, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=1000 + trial)
        s.init_pheromone(1.0)
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 1))
        s.solve(p, ids[0], ids[1], streams=[trial])
        ctx.check(ctx.lib.wa_acs_debug_counters(s.h, out.ctypes.data, 0))
        t = s.trace()
        L, lens = s.ants()
        res[mode] = (t["steps"].copy(), t["finite"].copy(), bits(t["bestL"]).copy(), bits(L).copy(), lens.copy(), bits(s.pheromone()).copy())
        assert int(out[9]) == int(out[7]) and (mode == "1" or int(out[9]) == 0)
        s.close()
    grid.close()
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)
