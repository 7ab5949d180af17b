"""Randomised differential sweep: seeded random small problems (grid shape, obstacle density, voxel pitch, alpha, beta,
rho, pheromone_0, colony rule, generations, hash size) solved by the dense solver, the lazily evaporating solver and
the 26-neighbour solver, each against the C oracle -- traces, best path and the full pheromone field bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import api

pytestmark = pytest.mark.gpu
CHUNKS = int(os.environ.get("WA_RANDOM_CHUNKS", "16"))   # 10 problems each; raise it for a soak run
CHUNKS4 = max(4, CHUNKS // 4)


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


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_random_problems_dense_lazy_and_26(ctx, chunk):
    rs = np.random.RandomState(1000 + chunk)
    for i in range(10):
        og, sid, eid, par, fixed, predict, iters, seed, stream = random_case(rs)
        hl = str(int(rs.choice([6, 8, 11])))
        os.environ["WA_HASH_LOG2"] = hl          # small tables: the bitmap spill path gets its share
        try:
            bound = fixed if fixed else int(0.35 * predict / float(og.precision))
            lazy_ok = bound <= 2048 and int(0.2 * bound) + 1 <= 64
            for nb, lazy in ((6, False), (6, True), (26, False), (26, True)):
                if lazy and not lazy_ok:
                    continue
                try:
                    check(ctx, og, sid, eid, par, fixed, predict, iters, seed, stream, nb, lazy)
                except AssertionError as e:
                    raise AssertionError("chunk %d case %d nb %d lazy %s dims %dx%dx%d par %s fixed %d predict %g iters %d hash %s: %s" % (
                        chunk, i, nb, lazy, og.nx, og.ny, og.nz, par, fixed, predict, iters, hl, e))
        finally:
            del os.environ["WA_HASH_LOG2"]


# ---------------------------------------------------------------- the other kernels, same idea
@pytest.mark.parametrize("chunk", range(CHUNKS4))
def test_random_meshes_voxelise_and_resolve(ctx, chunk):
    rs = np.random.RandomState(2000 + chunk)
    for i in range(6):
        nt = int(rs.randint(1, 40))
        scale = float(rs.choice([0.05, 1.0, 30.0]))
        v = (rs.uniform(-1, 1, (nt, 3, 3)) * scale).astype(np.float32)
        if rs.rand() < 0.5:
            v = v[:, :1] + (rs.uniform(-0.15, 0.15, (nt, 3, 3)) * scale).astype(np.float32)   # small triangles
        nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        with np.errstate(invalid="ignore", divide="ignore"):
            nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
        tris = np.zeros((nt, 12), np.float32)
        tris[:, :3] = nrm
        tris[:, 3:] = v.reshape(nt, 9)
        if i % 3 == 2:   # every third mesh goes through its ASCII form (read_STL.hpp:99-129: the normals come back as 0)
            import stl_text
            text = stl_text.ascii_stl_text(tris, name="m%d" % i)
            tris = api.stl_parse(text)
            assert np.array_equal(bits(tris), bits(O.stl_parse(text))) and not tris[:, :3].any() and len(tris) == nt
        ext = float(np.ptp(v.reshape(-1, 3), axis=0).max())
        p = max(ext / float(rs.randint(3, 40)), 1e-4 * scale)
        wall = int(rs.randint(0, 6))
        og = O.grid_from_mesh(tris, p, wall)
        if og.n > 400000:
            continue
        dg = api.Grid.from_mesh(ctx, tris, p, wall)
        assert (dg.nx, dg.ny, dg.nz) == (og.nx, og.ny, og.nz), (chunk, i)
        assert np.array_equal(dg.occupancy(), og.free), (chunk, i, nt, p, wall)
        pts = np.stack([rs.choice(og.cx, 12), rs.choice(og.cy, 12), rs.choice(og.cz, 12)], axis=1).astype(np.float32)
        pts += rs.uniform(-1.5 * p, 1.5 * p, pts.shape).astype(np.float32)
        want = np.array([og.resolve(q) for q in pts])
        assert np.array_equal(dg.resolve(pts), want), (chunk, i)                                   # 12 points: the host mirror of the grid
        assert np.array_equal(dg.resolve(np.concatenate([pts, pts[::-1]])), np.concatenate([want, want[::-1]])), (chunk, i)   # 24: the kernel
        dg.close()


@pytest.mark.parametrize("chunk", range(CHUNKS4))
def test_random_splines(ctx, chunk):
    rs = np.random.RandomState(3000 + chunk)
    for i in range(10):
        dim, deg = int(rs.randint(1, 8)), int(rs.randint(0, 8))
        ci, cf = int(rs.randint(0, deg + 1)), int(rs.randint(0, deg + 1))
        n = int(rs.randint(max(0, deg - ci - cf - 1), 400))
        if deg + n + 2 + ci + cf + 1 < 2 * (deg + 1):
            continue
        tf = float(np.float32(rs.choice([1.0, 150.0, 6000.0, 0.37])))
        mid = np.cumsum(rs.uniform(-0.05, 0.05, size=(n, dim)), axis=0).astype(np.float32)
        init = rs.uniform(-1, 1, size=(ci + 1, dim)).astype(np.float32)
        fin = rs.uniform(-1, 1, size=(cf + 1, dim)).astype(np.float32)
        fill = int(rs.choice([0, 0x3f800000, 0x7fc00000]))
        ob = O.Bspline(dim, deg, ci, cf, n, fill)
        ob.set_param(init, fin, mid, tf)
        b = api.Bspline(ctx, dim, deg, ci, cf, n, fill)
        b.set_param(init, fin, mid, tf)
        k, c = b.arrays()

        def canon(a):   # NaN payloads are not reproduced
            a = np.ascontiguousarray(a, np.float32)
            u = a.view(np.uint32).ravel().copy()
            u[np.isnan(a).ravel()] = 0x7fc00000
            return u
        assert np.array_equal(canon(k), canon(ob.knots)) and np.array_equal(canon(c), canon(ob.cps)), (chunk, i, dim, deg, ci, cf, n)
        us = rs.uniform(-0.1 * tf, 1.1 * tf, size=200).astype(np.float32)
        for der in sorted({0, min(1, deg), deg}):
            got, ok = b.eval(us, der)
            want, wok = ob.eval(us, der, prefill=0.0)
            assert np.array_equal(ok, wok) and np.array_equal(canon(got), canon(want)), (chunk, i, dim, deg, ci, cf, n, der)
        b.close()


@pytest.mark.parametrize("wave", ["1", "0"])
def test_random_seam_ordering(ctx, wave):
    rs = np.random.RandomState(4000)
    os.environ["WA_GTSP_WAVE"] = wave
    try:
        for i in range(16):
            n = int(rs.randint(2, 48))
            P = rs.uniform(0, 1, (n, 3))
            if rs.rand() < 0.3:
                P[rs.randint(n)] = P[rs.randint(n)]
            d = np.abs(P[:, None, :] - P[None, :, :]).sum(-1) if rs.rand() < 0.5 else np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
            d = np.round(d, 3) if rs.rand() < 0.5 else d        # graph.in carries 3 decimals: many exact ties
            cap = int(rs.choice([0, 0, 1, 7, 30]))
            seed, stream = int(rs.randint(1 << 30)), int(rs.randint(100))
            o = O.gtsp_solve(d, mode=O.DEV, seed=seed, stream=stream, max_iterations=cap, want_pher=True)
            t = api.gtsp_solve(ctx, d, mode=api.RNG_DEV, seed=seed, stream=stream, max_iterations=cap, want_pher=True)
            assert t["iters"][0] == o["iters"] and t["L"][0] == o["L"] and np.array_equal(t["edges"][0], o["edges"]), (i, n, cap)
            assert np.array_equal(t["pher"][0].view(np.uint64), o["pher"].view(np.uint64)), (i, n, cap)
    finally:
        del os.environ["WA_GTSP_WAVE"]


@pytest.mark.parametrize("chunk", range(max(3, CHUNKS // 8)))
def test_random_medium_problems(ctx, chunk):
    """larger grids and colonies: replay after convergence, > 64 depositing ranks (unfused chunks), > 2048 ants"""
    rs = np.random.RandomState(5000 + chunk)
    for i in range(5):
        n = int(rs.randint(14, 34))
        og = O.synth_grid(n, seed=int(rs.randint(1 << 20)), occ_prob=float(rs.choice([0.0, 0.1, 0.2])))
        sid, eid = og.resolve(np.zeros(3, np.float32)), og.resolve(np.full(3, n - 1, np.float32))
        par = dict(alpha=int(rs.choice([1, 1, 2])), beta=float(np.float32(rs.choice([0.6, 1.2]))),
                   rho=float(np.float32(rs.choice([0.8, 0.6]))), pheromone_0=1.0)
        fixed = int(rs.choice([30, 64, 200, 330, 2100] if i < 4 else [2100]))
        iters = int(rs.choice([8, 20, 45])) if fixed < 2000 else 3
        seed, stream = int(rs.randint(1 << 30)), int(rs.randint(9))
        lazy_ok = fixed <= 2048 and int(0.2 * fixed) + 1 <= 64
        for nb, lazy in ((6, False), (6, True), (26, False), (26, True)):
            if lazy and not lazy_ok:
                continue
            if nb == 26 and fixed > 400:
                continue
            try:
                check(ctx, og, sid, eid, par, fixed, 3.0 * n, iters, seed, stream, nb, lazy)
            except AssertionError as e:
                raise AssertionError("medium chunk %d case %d nb %d lazy %s n %d par %s fixed %d iters %d: %s" % (chunk, i, nb, lazy, n, par, fixed, iters, e))


@pytest.mark.parametrize("chunk", range(CHUNKS4))
def test_random_problems_ref_mode(ctx, chunk):
    """WA_RNG_REF (libc stream carried on the device, libstdc++ sort order) against the oracle's REF mode, which the
    live differential test pins to the reference itself: same draws consumed, same stream position afterwards."""
    rs = np.random.RandomState(6000 + chunk)
    for i in range(8):
        og, sid, eid, par, fixed, predict, iters, seed, stream = random_case(rs)
        iters = min(iters, 25)
        for nb in (6, 26):
            dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
            bound = fixed if fixed else int(0.35 * predict / float(og.precision))
            s = api.AcsSolver(ctx, dg, 1, max(bound, 1), neighbourhood=nb)
            s.init_pheromone(par["pheromone_0"])
            s.srand(seed & 0x7fffffff)
            p = api.default_params(max_iteration=iters, predict=predict, fixed_colony=fixed, rng_mode=api.RNG_REF, **par)
            s.solve(p, sid, eid)
            rng = O.srand(seed & 0x7fffffff)
            a = O.Acs(og, pheromone_0=par["pheromone_0"], nb=nb)
            tr = a.solve(sid, eid, iters, predict, fixed_colony=fixed, mode=O.REF, rng=rng, **par)
            t = s.trace()
            tag = (chunk, i, nb, og.nx, og.ny, og.nz, par, fixed, predict, iters)
            assert np.array_equal(t["steps"], tr["steps"]) and np.array_equal(bits(t["bestL"]), bits(tr["bestL"])), tag
            cost, path, ch = s.result()
            assert bits(cost) == bits(a.best_L), tag
            if np.isfinite(cost):
                assert np.array_equal(path, a.best_path()[0]), tag
            assert np.array_equal(bits(s.pheromone()), bits(a.pheromone())), tag
            st = s.rand_state()
            assert [int(v) for v in st[:31]] == list(rng.r)[:31] and int(st[34]) == rng.f and int(st[35]) == rng.b, tag   # 31 state words
            s.close()
            dg.close()
