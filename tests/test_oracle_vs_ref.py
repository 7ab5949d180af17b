"""Live differential: the C oracle (oracle/weld_oracle.c) against the REAL reference, i.e. oracle/_ref/ref_harness,
the reference's own headers compiled where they lie (oracle/Makefile) -- on seeded random inputs, beyond the
committed golden vectors.  CPU only.  Skipped where the harness binary is absent (it is git-ignored; `make -C oracle`
builds it in the container that has /root/reference, and it travels to the GPU box with the snapshot)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import waf
from tmpw import TMPW

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/ref_harness not built (needs /root/reference)")
TMP = TMPW + "weld_oracle_vs_ref_%d" % os.getuid()
os.makedirs(TMP, exist_ok=True)
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()


def random_unit_grid(rs):
    """wall 0, unit pitch, coords = index: enters the reference through GridMap::readGridMap exactly"""
    n = int(rs.randint(4, 13))
    free = (rs.uniform(size=n * n * n) >= rs.choice([0.0, 0.1, 0.25])).astype(np.uint8)
    f3 = free.reshape(n, n, n)
    f3[:2, :2, :2] = 1
    f3[n - 2:, n - 2:, n - 2:] = 1
    ax = np.arange(n, dtype=np.float32)
    return O.Grid(ax, ax.copy(), ax.copy(), free, 1.0, 0)


@pytest.mark.parametrize("chunk", range(4))
def test_acs_random_grids_both_neighbourhoods(chunk):
    rs = np.random.RandomState(100 + chunk)
    for i in range(4):
        og = random_unit_grid(rs)
        n = og.nx
        gin = TMP + "/g.in"
        O.write_grid_in(og, gin)
        nb = int(rs.choice([6, 26]))
        fixed = int(rs.choice([0, 7, 20]))
        predict = float(np.float32(rs.uniform(6, 40)))
        iters = int(rs.choice([1, 5, 20, 40]))
        seed = int(rs.randint(1, 1 << 30))
        kw = dict(gridin=gin, spt="0,0,0", ept="%d,%d,%d" % (n - 1, n - 1, n - 1), seed=seed, iters=iters,
                  predict=repr(predict), driven=1, nb=nb, dumppher=1)
        if fixed:
            kw["fixed"] = fixed
        r = O.run_ref("acs", TMP + "/a.waf", **kw)
        sid, eid = og.resolve(np.zeros(3, np.float32)), og.resolve(np.full(3, n - 1, np.float32))
        assert sid == waf.scalar(r, "start_id") and eid == waf.scalar(r, "end_id")
        rng = O.srand(seed)
        a = O.Acs(og, nb=nb)
        tr = a.solve(sid, eid, iters, float(np.float32(predict)), fixed_colony=fixed, mode=O.REF, rng=rng)
        tag = (chunk, i, n, nb, fixed, predict, iters, seed)
        assert bits(a.best_L)[0] == bits(r["best_L"])[0], tag
        ids, ch = a.best_path()
        assert np.array_equal(ids, r["best_path"]) and np.array_equal(ch, r["best_choice"]), tag
        assert np.array_equal(bits(a.pheromone()), bits(r["pher"])), tag          # the whole field, bit for bit
        assert np.array_equal(tr["colony"], r["tr_colony"]) and np.array_equal(tr["steps"], r["tr_steps"]), tag
        assert np.array_equal(bits(tr["bestL"]), bits(r["tr_bestL"])), tag
        assert rng.calls == waf.scalar(r, "rand_calls") and O.rand(rng) == waf.scalar(r, "next_rand"), tag


def test_gtsp_random_graphs():
    rs = np.random.RandomState(7)
    for i in range(8):
        n = int(rs.randint(3, 40))
        P = rs.randint(0, 60, size=(n, 3))
        path = TMP + "/graph.in"
        with open(path, "w") as f:
            f.write("%d %d\n" % (n, n * (n - 1) // 2))
            for a in range(n):
                for b in range(a + 1, n):
                    f.write("%.3f\n" % (np.abs(P[a] - P[b]).sum() / 37.0 + 0.001))
        seed = int(rs.randint(1, 1 << 30))
        r = O.run_ref("gtsp", TMP + "/t.waf", graph=path, seed=seed)
        d = r["gtsp_dis"].reshape(n, n)
        rng = O.srand(seed)
        t = O.gtsp_solve(d, mode=O.REF, rng=rng, want_pher=True)
        assert t["iters"] == waf.scalar(r, "gtsp_iters") and t["L"] == waf.scalar(r, "tour_L"), (i, n)
        assert np.array_equal(t["edges"].reshape(-1), r["tour_edges"]), (i, n)
        assert np.array_equal(t["pher"].reshape(-1).view(np.uint64), r["gtsp_pher"].view(np.uint64)), (i, n)
        assert O.rand(rng) == waf.scalar(r, "next_rand"), (i, n)


def test_voxelise_random_precisions():
    rs = np.random.RandomState(11)
    for stl, lo, hi in (("cubic.stl", 0.008, 0.05), ("simplified_piece.stl", 0.02, 0.06)):
        mesh = O.stl_parse(open(os.path.join(G, stl), "rb").read())
        for i in range(3):
            p = repr(float(np.float32(rs.uniform(lo, hi))))
            wall = int(rs.randint(0, 9))
            r = O.run_ref("voxelize", TMP + "/v.waf", stl=os.path.join(G, stl), p=p, wall=wall)
            og = O.grid_from_mesh(mesh, float(p), wall)
            assert [og.nx, og.ny, og.nz, wall] == r["dims"].tolist(), (stl, p, wall)
            assert np.array_equal(bits(og.cx), bits(r["cx"])) and np.array_equal(bits(og.cz), bits(r["cz"]))
            assert np.array_equal(og.free, r["free"]), (stl, p, wall)


def test_splines_random():
    rs = np.random.RandomState(13)
    combos = [(0, 0, 0), (1, 0, 0), (1, 1, 1), (2, 0, 0), (2, 1, 1), (2, 2, 2), (2, 2, 1), (3, 0, 0), (3, 1, 1), (3, 2, 2),
              (3, 3, 3), (3, 1, 2), (4, 3, 3), (5, 2, 2), (5, 4, 4)]        # what the harness instantiates
    for i in range(12):
        d, ci, cf = combos[int(rs.randint(len(combos)))]
        n = int(rs.randint(max(1, d - ci - cf), 120))
        if d + n + 2 + ci + cf + 1 < 2 * (d + 1):
            continue
        tf = repr(float(np.float32(rs.choice([1.0, 150.0, 6000.0, 2.5]))))
        fill = rs.choice(["0", "3f800000", "bf000000"])
        g = O.run_ref("bspline", TMP + "/b.waf", deg=d, ci=ci, cf=cf, n=n, seed=int(rs.randint(1, 10 ** 6)), tf=tf, fill=fill,
                      pad=int(rs.choice([0, 6])), t0="-1", dt=repr(float(tf) / 40), count=50)
        b = O.Bspline(3, d, ci, cf, n, int(fill, 16))
        b.set_param(g["init"], g["fin"], g["middle"].reshape(n, -1), float(tf))
        assert np.array_equal(bits(b.knots), bits(g["knots"])) and np.array_equal(bits(b.cps), bits(g["cps"])), (i, d, ci, cf, n)
        for k in range(d + 2):
            out, ok = b.eval(g["u"], k)
            assert np.array_equal(ok, g["ok%d" % k]) and np.array_equal(bits(out), bits(g["der%d" % k])), (i, d, ci, cf, n, k)


def test_ascii_stl_reader_on_mutated_files():
    """read_STL.hpp:99-129 against the oracle's restatement on text files with random damage: tokens dropped, swapped, replaced by
    junk, lines cut, numbers respelt.  Every file ends in a NUL byte (the reference reads on past the end of its buffer otherwise).
    A file on which the reference's loop does not end (it ends behind a "facet" token) must be the one the oracle refuses."""
    import subprocess
    import stl_text
    rs = np.random.RandomState(77)
    cubic = O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read())
    std = stl_text.ascii_stl_text(cubic[:5])
    junk = [b"facet", b"vertex", b"endsolid", b"1e", b".", b"-", b"+.e1", b"0x10", b"inf", b"1e400", b"-1e-400", b"12abc", b"\t", b"\n", b"\r\n", b"\v", b" ",
            b"0" * 600 + b"125", b"9" * 700 + b"e-690", b"1" + b"0" * 50, b"-." + b"0" * 520 + b"7e520"]   # numbers longer than any fixed buffer
    n_cmp = n_refused = 0
    for case in range(120):
        toks = std.replace(b"\n", b" \n ").split(b" ")
        for _ in range(int(rs.randint(1, 6))):
            k = int(rs.randint(0, len(toks)))
            what = int(rs.randint(0, 4))
            if what == 0:
                del toks[k]
            elif what == 1:
                toks[k] = junk[int(rs.randint(0, len(junk)))]
            elif what == 2:
                toks.insert(k, junk[int(rs.randint(0, len(junk)))])
            else:
                del toks[k:k + int(rs.randint(1, 30))]
        data = b" ".join(toks)
        if rs.uniform() < 0.3:
            data = data[:int(rs.randint(80, max(81, len(data))))]
        data = data.ljust(81, b" ") + b"\0"
        if case == 0:
            data = b"solid s\n" + b" " * 80 + b"facet\0"      # the text ends behind a "facet" token
        f = TMP + "/fuzz.stl"
        open(f, "wb").write(data)
        try:
            mine = O.stl_parse(data)
        except ValueError:
            mine = None
        try:
            subprocess.check_call([O.REF_BIN, "stl", "out=%s" % (TMP + "/fz.waf"), "stl=%s" % f], stderr=subprocess.DEVNULL, timeout=5,
                                  preexec_fn=lambda: __import__("resource").setrlimit(__import__("resource").RLIMIT_AS, (1 << 30, 1 << 30)))
            r = waf.load(TMP + "/fz.waf")
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError):
            assert mine is None, (case, data)       # the endless push_back: killed by the clock or by the address-space limit
            n_refused += 1
            continue
        assert mine is not None, (case, data)
        ref = np.asarray(r["tris"], np.float32).reshape(-1, 12)
        assert mine.shape == ref.shape and np.array_equal(bits(mine), bits(ref)), (case, data)
        n_cmp += 1
    assert n_cmp >= 100 and n_refused >= 1
    print("ascii stl fuzz: %d files equal, %d refused on both sides" % (n_cmp, n_refused))
