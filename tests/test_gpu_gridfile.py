"""GridMap's file hand-offs through the drop-in header on the GPU: the reference-compatible
writer reproduces the reference's own file byte for byte, the reader rebuilds from that file exactly what the
reference's reader rebuilds (Q5 included), the default format round-trips, a truncated file is an error."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import waf
from welding_robot_amd import _lib, build
from tmpw import TMPW

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = TMPW + "weldacs_gridfile_check_%d" % os.getuid()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "welding_robot_amd", "include"),
                        os.path.join(ROOT, "tests", "cpp", "gridfile_check.cpp"), "-L" + libdir, "-lweldacs", "-Wl,-rpath," + libdir, "-o", EXE],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def parse(path):
    d = {}
    for line in open(path):
        t = line.split()
        if t[0] == "status":
            d["status"] = int(t[1])
        elif t[0] == "dims":
            d["dims"] = [int(v) for v in t[1:]]
        elif t[0] == "precision":
            d["precision"] = np.array([int(t[1], 16)], np.uint32)
        elif t[0] in ("cx", "cy", "cz"):
            d[t[0]] = np.array([int(v, 16) for v in t[1:]], np.uint32)
        elif t[0] == "free":
            d["free"] = np.frombuffer(t[1].encode(), np.uint8) - ord("0")
    return d


def golden():
    g = waf.load(os.path.join(G, "gridfile_cubic.waf"))
    n = int(np.prod(g["made_dims"][:3]))
    return g, n


def test_compat_writer_reproduces_the_reference_file_byte_for_byte(exe, tmp_path):
    g, n = golden()
    f, d = str(tmp_path / "c.in"), str(tmp_path / "c.txt")
    r = subprocess.run([exe, "write", os.path.join(G, "cubic.stl"), "0.0219", "8", f, "1", d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(f, "rb").read() == g["file_text"].tobytes()
    m = parse(d)   # the grid in memory is the one the reference MADE (true bounding box)
    assert m["status"] == 0 and m["dims"] == g["made_dims"].tolist()
    assert np.array_equal(m["cx"], bits(g["made_cx"])) and np.array_equal(m["cy"], bits(g["made_cy"])) and np.array_equal(m["cz"], bits(g["made_cz"]))
    assert np.array_equal(m["free"], np.unpackbits(g["made_free_packed"])[:n])


def test_reader_on_the_reference_file_equals_the_reference_reader(exe, tmp_path):
    g, n = golden()
    f, d = str(tmp_path / "ref.in"), str(tmp_path / "ref.txt")
    open(f, "wb").write(g["file_text"].tobytes())
    assert subprocess.run([exe, "read", f, d], capture_output=True, text=True).returncode == 0
    m = parse(d)
    assert m["status"] == 0 and m["dims"] == g["read_dims"].tolist()
    assert np.array_equal(m["precision"], bits(g["read_precision"]))
    assert np.array_equal(m["cx"], bits(g["read_cx"])) and np.array_equal(m["cy"], bits(g["read_cy"])) and np.array_equal(m["cz"], bits(g["read_cz"]))
    assert np.array_equal(m["free"], np.unpackbits(g["read_free_packed"])[:n])


def test_default_format_round_trips_exactly(exe, tmp_path):
    g, n = golden()
    f, d1, d2 = str(tmp_path / "d.in"), str(tmp_path / "d1.txt"), str(tmp_path / "d2.txt")
    assert subprocess.run([exe, "write", os.path.join(G, "cubic.stl"), "0.0219", "8", f, "0", d1], capture_output=True).returncode == 0
    assert subprocess.run([exe, "read", f, d2], capture_output=True).returncode == 0
    a, b = parse(d1), parse(d2)
    assert a["dims"] == b["dims"] and np.array_equal(a["precision"], b["precision"]) and np.array_equal(a["free"], b["free"])
    for k in ("cx", "cy", "cz"):
        assert np.array_equal(a[k], b[k]), k      # true bounding box, 9 significant digits: the same floats come back
    # ... and the oracle's restatement of the REFERENCE reader makes the same grid of this file
    og, got = O.read_grid_in(f)
    assert got == n and np.array_equal(bits(og.cx), b["cx"]) and np.array_equal(og.free, b["free"])


def test_truncated_file_is_an_error_not_a_free_grid(exe, tmp_path):
    g, n = golden()
    text = g["file_text"].tobytes()
    f, d = str(tmp_path / "t.in"), str(tmp_path / "t.txt")
    open(f, "wb").write(text[:len(text) // 2])
    r = subprocess.run([exe, "read", f, d], capture_output=True, text=True)
    m = parse(d)
    assert m["status"] == 5 and "dims" not in m and "Truncated" in r.stdout     # WA_ERR_FORMAT, no grid
    open(f, "wb").write(b"20000 25 32\n")
    subprocess.run([exe, "read", f, d], capture_output=True, text=True)
    assert parse(d)["status"] == 5
    subprocess.run([exe, "read", str(tmp_path / "missing.in"), d], capture_output=True, text=True)
    assert parse(d)["status"] == 4                                              # WA_ERR_FILE


@pytest.mark.ref
@pytest.mark.skipif(not O.have_ref(), reason="compiled reference absent")
def test_reference_reader_accepts_the_default_format(exe, tmp_path):
    """drop-in writer -> the REAL reference's readGridMap (oracle/_ref/ref_harness travels with the snapshot)."""
    f, d = str(tmp_path / "x.in"), str(tmp_path / "x.txt")
    assert subprocess.run([exe, "write", os.path.join(G, "cubic.stl"), "0.0219", "8", f, "0", d], capture_output=True).returncode == 0
    m = parse(d)
    r = O.run_ref("voxelize", str(tmp_path / "x.waf"), gridin=f)
    assert r["dims"].tolist() == m["dims"] and np.array_equal(r["free"], m["free"])
    assert np.array_equal(bits(r["cx"]), m["cx"]) and np.array_equal(bits(r["cy"]), m["cy"]) and np.array_equal(bits(r["cz"]), m["cz"])
