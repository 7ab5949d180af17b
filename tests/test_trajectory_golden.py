"""Trajectory post-processing: the C oracle's restatement of BS_Basic
(core/BSplineBasic.h) and of ACS_GTSP::read_all_segments against golden vectors captured from
the REAL reference classes (tests/golden/make_golden.py bspline).  Bit-exact, CPU only."""
import os

import numpy as np
import pytest

import oracle_lib as O
import waf

G = os.path.join(os.path.dirname(__file__), "golden")
_cases = waf.load(os.path.join(G, "bspline_cases.waf"))
TAGS = sorted({k.split("/")[0] for k in _cases})


def case(tag):
    pre = tag + "/"
    return {k[len(pre):]: v for k, v in _cases.items() if k.startswith(pre)}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()


def oracle_spline(g, dim=3):
    n = waf.scalar(g, "n_middle")
    b = O.Bspline(dim, waf.scalar(g, "deg"), waf.scalar(g, "ci"), waf.scalar(g, "cf"), n,
                  waf.scalar(g, "fill_bits"))
    b.set_param(g["init"], g["fin"], g["middle"].reshape(n, -1), waf.scalar(g, "tf"))
    return b


@pytest.mark.parametrize("tag", TAGS)
def test_bspline_golden(tag):
    g = case(tag)
    b = oracle_spline(g)
    assert np.array_equal(bits(b.knots), bits(g["knots"]))
    assert np.array_equal(bits(b.cps), bits(g["cps"]))
    for d in range(b.degree + 2):          # d = degree+1: getCurveDerPoint refuses, output untouched
        out, ok = b.eval(g["u"], d)
        assert np.array_equal(ok, g["ok%d" % d]), d
        assert np.array_equal(bits(out), bits(g["der%d" % d])), d
    out, ok = b.sample(waf.scalar(g, "t0"), waf.scalar(g, "dt"), waf.scalar(g, "count"))
    assert np.array_equal(bits(out), bits(g["samples"]))
    assert ok.all()


def test_bspline_uninitialised_cell_is_an_input():
    """BS_Basic<float,3,2,2,2> (main.cpp:337) reads c_mat[idx][3], which _BasisFunsDers never
    writes (BSplineBasic.h:414-431): only the two constrained control points before the last one
    depend on it, and the restatement reproduces the reference for every fill pattern."""
    a, b = case("d2_main2"), case("d2_main2_fill1")
    assert np.array_equal(bits(a["middle"]), bits(b["middle"]))
    ca, cb = a["cps"].reshape(-1, 3), b["cps"].reshape(-1, 3)
    diff = np.flatnonzero((bits(ca) != bits(cb)).reshape(-1, 3).any(axis=1))
    assert diff.tolist() == [len(ca) - 3, len(ca) - 2]
    assert np.isnan(case("d2_main2_fillnan")["cps"].reshape(-1, 3)[-2]).all()
    # DEGREE = CL+1 never touches the uninitialised column
    g = case("d3")
    b1 = oracle_spline(g)
    g2 = dict(g)
    g2["fill_bits"] = np.array([0x7fc00000], np.int64)
    b2 = oracle_spline(g2)
    assert np.array_equal(bits(b1.cps), bits(b2.cps))


def test_bspline_invalid_arguments():
    with pytest.raises(ValueError):
        O.Bspline(3, 2, 3, 0, 10)          # constraint level above DEGREE: reference indexes ndu[-1]
    with pytest.raises(ValueError):
        O.Bspline(3, 8, 0, 0, 10)
    with pytest.raises(ValueError):
        O.Bspline(3, 5, 0, 0, 1)           # NumKnots < 2*(DEGREE+1): "Invalid setup" (:53-55)
    b = O.Bspline(3, 0, 0, 0, 0)           # no middle points at all is legal
    b.set_param([1, 2, 3], [4, 5, 6], np.zeros((0, 3), np.float32), 10.0)
    out, ok = b.eval([0.0, 4.9, 5.0, 10.0, 11.0])
    assert ok.all() and out[:2].tolist() == [[1, 2, 3]] * 2 and out[2:].tolist() == [[4, 5, 6]] * 3


def test_bspline_degree0_is_a_time_indexed_lookup():
    """BS_Basic<float,3,0,0,0> (main.cpp:299): knot span i holds control point i -- init, the path
    points, fin -- so pass 1 of main.cpp is a resampling of the stitched path."""
    g = case("d0_main1")
    b = oracle_spline(g)
    n = waf.scalar(g, "n_middle")
    k = b.knots
    mids = 0.5 * (k[:-1].astype(np.float64) + k[1:])
    out, ok = b.eval(mids.astype(np.float32))
    assert ok.all()
    want = np.vstack([g["init"][:3], g["middle"].reshape(n, -1)[:, :3], g["fin"][:3]])
    assert np.array_equal(bits(out), bits(want))


def _cubic_grid():
    return O.grid_from_mesh(O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read()), float("0.0219"), 8)


def segments_of(g, P=5):
    """(seg_ids, seg_off) of the first N-1 tour edges, as read_all_segments walks them."""
    lens = g["pair_len"].reshape(P, P)
    paths, off = {}, 0
    for i in range(P):
        for j in range(i + 1, P):
            paths[(i, j)] = paths[(j, i)] = g["pair_paths_upper"][off:off + lens[i, j]]
            off += lens[i, j]
    edges = g["tour_edges"].reshape(-1, 2)[:-1]
    segs = [paths[(int(a), int(b))].astype(np.int64) for a, b in edges]
    return np.concatenate(segs), np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)


@pytest.mark.parametrize("fill", ["0", "3f800000"])
def test_stitch_and_two_pass_smoothing_golden(fill):
    """main.cpp:283-352 on the cubic demo: stitched path, BS_Basic<3,0,0,0> over 150 ticks sampled
    every 10, BS_Basic<3,2,2,2> over 6000 ticks sampled every 50."""
    g = waf.load(os.path.join(G, "smooth_cubic_fill%s.waf" % fill))
    ref = waf.load(os.path.join(G, "pairs_cubic.waf"))
    assert np.array_equal(g["g_path_x"], ref["g_path_x"])        # same run as the pair-flow golden
    grid = _cubic_grid()
    ids, off = segments_of(g)
    assert len(off) - 1 == waf.scalar(g, "segments")
    xyz = O.stitch_segments(ids, off, None, grid.nx, grid.ny, grid.cx, grid.cy, grid.cz)
    want = np.stack([g["g_path_x"], g["g_path_y"], g["g_path_z"]], axis=1)
    assert np.array_equal(bits(xyz), bits(want))
    n = len(xyz)
    s1 = O.Bspline(3, 0, 0, 0, n, int(fill, 16))
    s1.set_param(xyz[0], xyz[-1], xyz, 150.0)
    assert np.array_equal(bits(s1.knots), bits(g["s1_knots"]))
    p1, _ = s1.sample(10.0, 10.0, 16)
    assert np.array_equal(bits(p1), bits(g["s1_samples"]))
    m2 = np.full((16, 9), 0.05, np.float32)
    m2[:, :3] = p1
    assert np.array_equal(bits(m2), bits(g["s2_middle"]))
    s2 = O.Bspline(3, 2, 2, 2, 16, int(fill, 16))
    s2.set_param(g["s2_init"], g["s2_fin"], m2, 6000.0)
    assert np.array_equal(bits(s2.cps), bits(g["s2_cps"]))
    p2, _ = s2.sample(50.0, 50.0, 121)
    assert np.array_equal(bits(p2), bits(g["s2_samples"]))


def test_stitch_reversal_option():
    """read_all_segments appends best_matrix[i][j] as stored even when the tour runs j -> i
    (ACS_GTSP.hpp:288-297); the reversal flag is the fix SURVEY 8(f) N3 asks for."""
    g = waf.load(os.path.join(G, "smooth_cubic_fill0.waf"))
    grid = _cubic_grid()
    ids, off = segments_of(g)
    edges = g["tour_edges"].reshape(-1, 2)[:-1]
    rev = (edges[:, 0] > edges[:, 1]).astype(np.uint8)
    assert rev.any()
    plain = O.stitch_segments(ids, off, None, grid.nx, grid.ny, grid.cx, grid.cy, grid.cz)
    fixed = O.stitch_segments(ids, off, rev, grid.nx, grid.ny, grid.cx, grid.cy, grid.cz)
    for s in range(len(rev)):
        a, b = plain[off[s]:off[s + 1]], fixed[off[s]:off[s + 1]]
        assert np.array_equal(b, a[::-1] if rev[s] else a)
    # with reversal every segment starts where the previous one ended
    for s in range(1, len(rev)):
        assert np.array_equal(fixed[off[s]], fixed[off[s] - 1])
