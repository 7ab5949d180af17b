"""GPU parity for the path post-processing kernels: k_stitch, k_bspline_setup,
k_bspline_middle, k_bspline_eval through the C ABI, against
  (1) golden vectors captured from the REAL BS_Basic / read_all_segments (tests/golden/make_golden.py),
  (2) the C oracle on larger seeded inputs.
fp32 bit patterns must be identical (NaN payloads excepted)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import waf
from test_trajectory_golden import TAGS, _cubic_grid, case, segments_of
from welding_robot_amd import api
from welding_robot_amd._lib import WeldacsError

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    """fp32 bit patterns; NaNs compare equal whatever their sign/payload (x86's default NaN is 0xffc00000,
    gfx950's 0x7fc00000, and source-negation modifiers flip a NaN's sign bit)."""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).ravel().copy()
    b[np.isnan(a).ravel()] = 0x7fc00000
    return b


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def same_where_ok(out, ok, want, want_ok):
    assert np.array_equal(ok, want_ok)
    m = ok.astype(bool)
    assert np.array_equal(bits(out[m]), bits(np.asarray(want).reshape(len(ok), -1)[m]))
    assert not out[~m].any()          # refused rows are zero-filled by the ABI


@pytest.mark.parametrize("tag", TAGS)
def test_bspline_matches_reference(ctx, tag):
    g = case(tag)
    n, deg = waf.scalar(g, "n_middle"), waf.scalar(g, "deg")
    b = api.Bspline(ctx, 3, deg, waf.scalar(g, "ci"), waf.scalar(g, "cf"), n, waf.scalar(g, "fill_bits"))
    b.set_param(g["init"], g["fin"], g["middle"].reshape(n, -1), waf.scalar(g, "tf"))
    knots, cps = b.arrays()
    assert np.array_equal(bits(knots), bits(g["knots"]))
    assert np.array_equal(bits(cps), bits(g["cps"]))
    for d in range(deg + 2):
        out, ok = b.eval(g["u"], d)
        same_where_ok(out, ok, g["der%d" % d], g["ok%d" % d])
    out, ok = b.sample(waf.scalar(g, "t0"), waf.scalar(g, "dt"), waf.scalar(g, "count"))
    assert ok.all() and np.array_equal(bits(out), bits(g["samples"]))
    b.close()


@pytest.mark.parametrize("fill", ["0", "3f800000"])
def test_main_flow_device_resident(ctx, fill):
    """main.cpp:283-352: stitched path -> BS_Basic<3,0,0,0> -> BS_Basic<3,2,2,2>, every intermediate kept
    on the device (Trajectory handles), against the reference's own run of that sequence."""
    g = waf.load(os.path.join(G, "smooth_cubic_fill%s.waf" % fill))
    og = _cubic_grid()
    grid = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    ids, off = segments_of(g)
    segs = [ids[off[s]:off[s + 1]] for s in range(len(off) - 1)]
    path = api.Trajectory.stitch(grid, segs)
    xyz = path.points()
    want = np.stack([g["g_path_x"], g["g_path_y"], g["g_path_z"]], axis=1)
    assert np.array_equal(bits(xyz), bits(want))
    s1 = api.Bspline(ctx, 3, 0, 0, 0, len(path), int(fill, 16))
    s1.set_param(xyz[0], xyz[-1], path, 150.0)
    p1, ok1, t1 = s1.sample(10.0, 10.0, 16, device=True)
    assert np.array_equal(bits(p1), bits(g["s1_samples"]))
    s2 = api.Bspline(ctx, 3, 2, 2, 2, len(t1), int(fill, 16))
    s2.set_param(g["s2_init"], g["s2_fin"], t1, 6000.0)
    k2, c2 = s2.arrays()
    assert np.array_equal(bits(k2), bits(g["s2_knots"])) and np.array_equal(bits(c2), bits(g["s2_cps"]))
    p2, ok2 = s2.sample(50.0, 50.0, 121)
    assert ok2.all() and np.array_equal(bits(p2), bits(g["s2_samples"]))
    # reversal option against the oracle
    edges = g["tour_edges"].reshape(-1, 2)[:-1]
    rev = (edges[:, 0] > edges[:, 1]).astype(np.uint8)
    fixed = api.Trajectory.stitch(grid, segs, rev).points()
    assert np.array_equal(bits(fixed), bits(O.stitch_segments(ids, off, rev, og.nx, og.ny, og.cx, og.cy, og.cz)))


@pytest.mark.parametrize("dim,deg,ci,cf,n,tf,count", [(3, 3, 2, 2, 100000, 6000.0, 1 << 20), (6, 5, 4, 4, 4096, 10.0, 50000),
                                                      (1, 1, 0, 0, 3, 1.0, 1000), (16, 7, 7, 7, 2000, 33.0, 20000),
                                                      (3, 0, 0, 0, 250000, 150.0, 300000), (2, 2, 1, 0, 777, 5.5, 4097)])
def test_bspline_large_vs_oracle(ctx, dim, deg, ci, cf, n, tf, count):
    rs = np.random.RandomState(dim * 1000 + deg)
    mid = np.cumsum(rs.uniform(-0.01, 0.01, size=(n, dim)), axis=0).astype(np.float32)
    init = rs.uniform(-1, 1, size=(ci + 1, dim)).astype(np.float32)
    fin = rs.uniform(-1, 1, size=(cf + 1, dim)).astype(np.float32)
    ob = O.Bspline(dim, deg, ci, cf, n)
    ob.set_param(init, fin, mid, tf)
    b = api.Bspline(ctx, dim, deg, ci, cf, n)
    b.set_param(init, fin, mid, tf)
    knots, cps = b.arrays()
    assert np.array_equal(bits(knots), bits(ob.knots))          # the sequential fp32 knot chain, drift included
    assert np.array_equal(bits(cps), bits(ob.cps))
    t0, dt = np.float32(-0.01 * tf), np.float32(1.03 * tf / count)
    for der in sorted({0, min(1, deg), deg}):
        got, ok = b.sample(t0, dt, count, der)
        want, wok = ob.sample(t0, dt, count, der, prefill=0.0)
        assert np.array_equal(ok, wok)
        assert np.array_equal(bits(got), bits(want)), der
    us = rs.uniform(-1, tf + 1, size=5000).astype(np.float32)
    got, ok = b.eval(us)
    want, wok = ob.eval(us, prefill=0.0)
    assert np.array_equal(ok, wok) and np.array_equal(bits(got), bits(want))
    b.close()


def test_knot_chain_drift_is_reproduced(ctx):
    """With 250k control points the fp32 recurrence K[i] = K[i-1] + step drifts visibly from i*step;
    the device reproduces the recurrence, not the closed form."""
    n = 250000
    b = api.Bspline(ctx, 3, 0, 0, 0, n)
    b.set_param([0, 0, 0], [1, 1, 1], np.zeros((n, 3), np.float32), 150.0)
    k, _ = b.arrays()
    step = np.float32(150.0) / np.float32(n + 2)
    closed = (np.arange(1, n + 2, dtype=np.float64) * float(step)).astype(np.float32)
    assert not np.array_equal(k[1:n + 2], closed)
    ob = O.Bspline(3, 0, 0, 0, n)
    ob.set_param([0, 0, 0], [1, 1, 1], np.zeros((n, 3), np.float32), 150.0)
    assert np.array_equal(bits(k), bits(ob.knots))


def test_trajectory_error_codes(ctx):
    for args in [(0, 3, 0, 0, 5), (17, 3, 0, 0, 5), (3, 8, 0, 0, 5), (3, 2, 3, 0, 5), (3, 2, 0, 3, 5), (3, 5, 0, 0, 1), (3, 1, 0, 0, -1)]:
        with pytest.raises(WeldacsError) as e:
            api.Bspline(ctx, *args)
        assert e.value.code == 1
    b = api.Bspline(ctx, 3, 2, 1, 1, 4)
    with pytest.raises(WeldacsError) as e:
        b.sample(0.0, 1.0, 4)
    assert e.value.code == 8                      # WA_ERR_STATE: SetParam has not run
    mid = np.zeros((4, 3), np.float32)
    for tf in (0.0, -1.0, np.inf, np.nan):
        with pytest.raises(WeldacsError) as e:
            b.set_param(np.zeros(6), np.zeros(6), mid, tf)
        assert e.value.code == 1
    b.set_param(np.zeros(6), np.ones(6), mid, 2.0)
    out, ok = b.eval([0.5], der=3)                # d > DEGREE: getCurveDerPoint returns false
    assert ok.tolist() == [0] and not out.any()
    out, ok = b.sample(0.0, 1.0, 0)
    assert out.shape == (0, 3)
    t = api.Trajectory.from_points(ctx, np.zeros((5, 3), np.float32))
    with pytest.raises(WeldacsError) as e:
        b.set_param(np.zeros(6), np.ones(6), t, 2.0)      # 5 points for n_middle = 4
    assert e.value.code == 1
    og = O.synth_grid(8, seed=1)
    grid = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, 0)
    with pytest.raises(WeldacsError) as e:
        api.Trajectory.stitch(grid, [[0, 1, 512]])         # node id outside the 8^3 grid
    assert e.value.code == 1
    empty = api.Trajectory.stitch(grid, [])
    assert len(empty) == 0 and empty.points().shape == (0, 3)
    ragged = api.Trajectory.stitch(grid, [[5], [], [7, 6]], [0, 1, 1])
    assert np.array_equal(ragged.points(), O.stitch_segments([5, 7, 6], [0, 1, 1, 3], [0, 1, 1], 8, 8, og.cx, og.cy, og.cz))
