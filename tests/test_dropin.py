"""The drop-in C++ headers (welding_robot_amd/include/core/*.hpp): same class names and call
sequence as the reference's main.cpp:273-283, compiled with the host g++ exactly like the reference
(-std=c++14, no hipcc, no Python.h) and linked against libweldacs.so.
CPU: the headers compile and the program refuses to run without a GPU.
GPU: REF mode reproduces the reference's own run of the same pipeline (golden pairs_cubic.waf)."""
import os
import subprocess

import numpy as np
import pytest

import waf
from welding_robot_amd import _lib, build
from tmpw import TMPW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = TMPW + "weldacs_dropin_demo_%d" % os.getuid()


def compile_demo():
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "welding_robot_amd", "include"), os.path.join(ROOT, "examples", "dropin_demo.cpp"),
           "-L" + libdir, "-lweldacs", "-Wl,-rpath," + libdir, "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr
    return EXE


def run_demo(mode, seed, out, stl="cubic.stl"):
    args = [EXE, os.path.join(G, stl), "0.0219", "8", os.path.join(G, "cubic_weld_points.in"), "0.5",
            out + ".graph", mode, str(seed), out]
    return subprocess.run(args, capture_output=True, text=True)


def parse(out):
    d = dict(cost={}, edges=[], gpath=[], smooth1=[], smooth2=[])
    cur = "gpath"
    for line in open(out):
        t = line.split()
        if t[0] in ("gpath", "smooth1", "smooth2"):
            cur = t[0]
        elif t[0] == "cost":
            d["cost"][(int(t[1]), int(t[2]))] = np.float32(t[3])
        elif t[0] == "tour_L":
            d["tour_L"], d["iters"] = float(t[1]), int(t[3])
        elif t[0] == "edge":
            d["edges"] += [int(t[1]), int(t[2])]
        elif t[0] != "points":
            d[cur].append([np.float32(v) for v in t])
    for k in ("gpath", "smooth1", "smooth2"):
        d[k] = np.array(d[k], np.float32).reshape(-1, 3)
    return d


def test_dropin_headers_compile_with_host_compiler_and_need_a_gpu():
    compile_demo()
    import ctypes as C
    h = C.c_void_p()
    lib = _lib.load()
    if lib.wa_ctx_create(0, C.byref(h)) == 0:
        lib.wa_ctx_destroy(h)
        pytest.skip("a HIP device is present")
    r = run_demo("dev", 1, TMPW + "weldacs_dropin_cpu.txt")
    assert r.returncode != 0 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("stl", ["cubic.stl", "cubic_ascii.stl"])
def test_dropin_pipeline_ref_mode_equals_reference(stl):
    """(cubic_ascii.stl: the same twelve triangles as text, through the ASCII branch of STLReader::readFile, read_STL.hpp:99-129.  Its
    normals stay 0 -- but the cube's faces are axis-aligned, so the reference's grid is the same one (the two goldens below), and with
    it every number of the pipeline.)"""
    if stl != "cubic.stl":
        a, b = waf.load(os.path.join(G, "vox_cubic_ascii_p0219_w8.waf")), waf.load(os.path.join(G, "vox_cubic_p0219_w8.waf"))
        assert all(np.array_equal(a[k], b[k]) for k in ("dims", "cx", "cy", "cz", "free_packed"))
    compile_demo()
    out = TMPW + "weldacs_dropin_ref.txt"
    r = run_demo("ref", 4321, out, stl)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = parse(out)
    g = waf.load(os.path.join(G, "pairs_cubic.waf"))
    P = 5
    want = g["pair_cost"].reshape(P, P)
    for i in range(P):
        for j in range(P):
            assert np.float32(d["cost"][(i, j)]).view(np.uint32) == want[i, j].view(np.uint32), (i, j)
    assert open(out + ".graph", "rb").read() == g["graph_text"].tobytes()  # incl. the Q6 header damage
    assert d["edges"] == g["tour_edges"].tolist() and d["tour_L"] == waf.scalar(g, "tour_L") and d["iters"] == waf.scalar(g, "gtsp_iters")
    assert np.array_equal(d["gpath"][:, 0].view(np.uint32), g["g_path_x"].view(np.uint32))
    assert np.array_equal(d["gpath"][:, 1].view(np.uint32), g["g_path_y"].view(np.uint32))
    assert np.array_equal(d["gpath"][:, 2].view(np.uint32), g["g_path_z"].view(np.uint32))
    # main.cpp:287-352 through the drop-in BS_Basic: the reference's own two smoothing passes (fixed times)
    sm = waf.load(os.path.join(G, "smooth_cubic_fill0.waf"))
    assert np.array_equal(d["smooth1"].view(np.uint32).ravel(), sm["s1_samples"].view(np.uint32))
    assert np.array_equal(d["smooth2"].view(np.uint32).ravel(), sm["s2_samples"].view(np.uint32))


@pytest.mark.gpu
def test_dropin_pipeline_dev_mode_is_deterministic_and_sane():
    compile_demo()
    a, b = TMPW + "weldacs_dropin_dev_a.txt", TMPW + "weldacs_dropin_dev_b.txt"
    assert run_demo("dev", 7, a).returncode == 0 and run_demo("dev", 7, b).returncode == 0
    da, db = parse(a), parse(b)
    assert da["edges"] == db["edges"] and da["tour_L"] == db["tour_L"] and np.array_equal(da["gpath"], db["gpath"])
    assert sorted(da["edges"][::2]) == [0, 1, 2, 3, 4]  # a Hamiltonian tour over the 5 weld points
    costs = np.array([da["cost"][(i, j)] for i in range(5) for j in range(5) if i != j])
    assert np.all(np.isfinite(costs)) and np.all(costs > 0)
    # lazy evaporation (the DEV default) and the dense sweep give the same bytes
    c = TMPW + "weldacs_dropin_dev_c.txt"
    assert run_demo("dev-dense", 7, c).returncode == 0
    assert open(a, "rb").read() == open(c, "rb").read()
    # the correct (non-compat) graph file parses back to the in-memory matrix
    tok = open(a + ".graph").read().split()
    assert tok[:2] == ["5", "10"] and np.float32(tok[2]) == da["cost"][(0, 1)]


@pytest.mark.gpu
def test_pair_loop_sharded_over_contexts_is_shard_invariant():
    """searchBestPathOfPoints' pair loop (ACSRank_3D.hpp:472-499) on one, two and three contexts -- host threads, a
    wa_ctx + grid replica + solver each, pairs dealt longest-first to the least loaded shard, stream key = global pair index
    -- gives the same costs and the same paths, bit for bit, with three slots per shard or with the slot count sized by
    rule (free memory, whole batches).  (One GPU here: an ordinal listed twice means two contexts on it.)"""
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = TMPW + "weldacs_shard_check_%d" % os.getuid()
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "welding_robot_amd", "include"),
                        os.path.join(ROOT, "tests", "cpp", "shard_check.cpp"), "-L" + libdir, "-lweldacs", "-Wl,-rpath," + libdir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr
    outs, shards = {}, {}
    for devs, slots in (("0", 3), ("0,0", 3), ("0,0,0", 3), ("all", 3), ("0", 0), ("0,0,0", 0), ("0,0,0,0,0,0,0,0", 0)):
        out = TMPW + "weldacs_shard_%s_%d.txt" % (devs.replace(",", "_"), slots)
        rr = subprocess.run([exe, os.path.join(G, "cubic.stl"), "0.0219", "8", os.path.join(G, "cubic_weld_points.in"), "0.5", "99", devs, out, str(slots)],
                            capture_output=True, text=True)
        assert rr.returncode == 0, rr.stdout[-1500:] + rr.stderr[-1500:]
        outs[(devs, slots)] = open(out, "rb").read()
        shards[(devs, slots)] = [[int(v) for v in l.split()[1:]] for l in open(out + ".shards")]
    ref = outs[("0", 3)]
    assert ref.count(b"pair ") == 10 and b"7f800000" not in ref      # 10 finite pair costs
    for k, v in outs.items():
        assert v == ref, k                                             # 1, 2, 3, 8 contexts; 3 slots or sized by rule
    # dealing: every pair exactly once, longest-first to the least loaded shard => weights within one search of each other
    for k, sh in shards.items():
        assert sum(r[2] for r in sh) == 10, (k, sh)
        assert all(r[4] == -(-r[2] // r[3]) for r in sh), (k, sh)     # batches = ceil(pairs / slots)
    one = shards[("0", 0)][0]
    assert one[2] == 10 and one[3] == 10 and one[4] == 1               # sized by rule: all ten searches in ONE batch
    three = shards[("0,0,0", 0)]
    assert sorted(r[2] for r in three) == [3, 3, 4] and all(r[4] == 1 for r in three)
    w = [r[5] for r in three]
    longest = max(r[5] for r in shards[("0,0,0,0,0,0,0,0", 0)])      # eight shards, ten searches: the heaviest shard holds 1-2 searches
    assert max(w) - min(w) <= longest
    assert len(shards[("0,0,0,0,0,0,0,0", 0)]) == 8


@pytest.mark.gpu
def test_pair_loop_called_again_reuses_what_its_contexts_kept():
    """searchBestPathOfPoints three times on one object, three shards on one GPU: the shards' contexts live on between the calls and keep the
    device blocks of the solvers each call destroys (wa_ctx_cached_bytes), so the second and third call allocate nothing new -- what is kept
    stops growing -- and every call's costs and paths are those of a single call; trimDeviceMemory() hands everything back."""
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = TMPW + "weldacs_shard_check_%d" % os.getuid()
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "welding_robot_amd", "include"),
                        os.path.join(ROOT, "tests", "cpp", "shard_check.cpp"), "-L" + libdir, "-lweldacs", "-Wl,-rpath," + libdir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr
    outs = {}
    for calls in (1, 3):
        out = TMPW + "weldacs_again_%d.txt" % calls
        rr = subprocess.run([exe, os.path.join(G, "cubic.stl"), "0.0219", "8", os.path.join(G, "cubic_weld_points.in"), "0.5", "99", "0,0,0", out, "0", str(calls)],
                            capture_output=True, text=True)
        assert rr.returncode == 0, rr.stdout[-1500:] + rr.stderr[-1500:]
        outs[calls] = open(out, "rb").read()
        rows = [l.split() for l in open(out + ".cache")]
        kept = [[int(v) for v in r_[2:]] for r_ in rows if r_[0] == "call"]
        assert len(kept) == calls and all(len(k) == 3 for k in kept)            # primary + two further shard contexts
        assert all(b > 0 for b in kept[0][1:])                                   # the further shards' solvers were destroyed: their blocks are kept
        for k in kept[1:]:
            assert k[1:] == kept[0][1:], kept                                    # ... and taken and given back by every further call: nothing new
        assert [int(v) for v in rows[-1][2:]] == [0, 0, 0] and rows[-1][0] == "trimmed"
    assert outs[3] == outs[1] and outs[1].count(b"pair ") == 10


@pytest.mark.gpu
def test_pair_loop_deals_whole_end_point_groups_when_there_are_enough():
    """Ten weld points = 45 pair searches, 9 end-point groups: with two contexts (9 >= 4 x 2) the C++ pair loop deals WHOLE groups
    longest-first (every end point lives on one shard), with one context nothing is dealt; costs and paths are the same bytes."""
    import oracle_lib as O
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = TMPW + "weldacs_shard_check_%d" % os.getuid()
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "welding_robot_amd", "include"),
                        os.path.join(ROOT, "tests", "cpp", "shard_check.cpp"), "-L" + libdir, "-lweldacs", "-Wl,-rpath," + libdir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr
    tris = O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read())
    og = O.grid_from_mesh(tris, np.float32("0.0219"), 8)
    nodes = [(4, 4, 4), (20, 27, 20), (4, 27, 20), (20, 4, 4), (12, 2, 12), (4, 4, 20), (20, 27, 4), (12, 29, 12), (2, 15, 12), (22, 15, 12)]
    pts = TMPW + "weldacs_ten_points_%d.in" % os.getpid()
    with open(pts, "w") as f:
        f.write("%d\n" % len(nodes))
        for z, y, x in nodes:
            assert og.free[(z * og.ny + y) * og.nx + x]
            f.write("%f %f %f\n" % tuple(og.node_pt(z, y, x)))
    outs, shards = {}, {}
    for devs in ("0", "0,0"):
        out = TMPW + "weldacs_groups_%s.txt" % devs.replace(",", "_")
        rr = subprocess.run([exe, os.path.join(G, "cubic.stl"), "0.0219", "8", pts, "0.5", "99", devs, out, "0"], capture_output=True, text=True)
        assert rr.returncode == 0, rr.stdout[-1500:] + rr.stderr[-1500:]
        outs[devs] = open(out, "rb").read()
        shards[devs] = [[int(v) for v in l.split()[1:]] for l in open(out + ".shards")]
        lines = [l for l in rr.stdout.splitlines() if l.startswith("[ACS 3D] shard")]
        assert len(lines) == len(devs.split(","))
    assert outs["0"].count(b"pair ") == 45 and outs["0"] == outs["0,0"]
    two = shards["0,0"]
    assert sum(r[2] for r in two) == 45 and all(r[2] > 0 for r in two)
    w = [r[5] for r in two]
    assert abs(w[0] - w[1]) <= 0.25 * max(w)      # whole groups of 1..9 searches: balanced to the size of a group
