"""The grid-map text cache (model_grid_map.hpp:275-294 writer, :300-356 reader) -- CPU side:
the oracle's restatement of both against the file the REFERENCE wrote for cubic.stl and against what the reference's
own reader rebuilt from it (golden gridfile_cubic.waf, made by tests/golden/make_golden.py gridfile)."""
import os

import numpy as np

import oracle_lib as O
import waf

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load():
    g = waf.load(os.path.join(G, "gridfile_cubic.waf"))
    tris = O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read())
    return g, tris


def test_writer_restated_is_byte_identical_to_the_reference_file(tmp_path):
    g, tris = load()
    og = O.grid_from_mesh(tris, 0.0219, 8)
    lo, hi = O.q5_bbox(tris, 0.0219)          # the last triangle's box, not the mesh's (Q5)
    f = str(tmp_path / "w.in")
    O.write_grid_in(og, f, lo, hi)
    assert open(f, "rb").read() == g["file_text"].tobytes()
    n = int(np.prod(g["made_dims"][:3]))
    assert np.array_equal(og.free, np.unpackbits(g["made_free_packed"])[:n])
    assert np.array_equal(bits(og.cx), bits(g["made_cx"])) and np.array_equal(bits(og.cz), bits(g["made_cz"]))


def test_reader_restated_equals_the_reference_reader_on_the_reference_file(tmp_path):
    g, _ = load()
    f = str(tmp_path / "r.in")
    open(f, "wb").write(g["file_text"].tobytes())
    og, got = O.read_grid_in(f)
    n = int(np.prod(g["read_dims"][:3]))
    assert got == n and [og.nx, og.ny, og.nz, og.wall] == g["read_dims"].tolist()
    assert bits(og.precision) == bits(g["read_precision"])[0]
    for a, k in ((og.cx, "read_cx"), (og.cy, "read_cy"), (og.cz, "read_cz")):
        assert np.array_equal(bits(a), bits(g[k])), k
    assert np.array_equal(og.free, np.unpackbits(g["read_free_packed"])[:n])
    # Q5: the file does not round-trip -- same occupancy, different coordinates (header = last triangle's box, "%f")
    assert np.array_equal(g["read_free_packed"], g["made_free_packed"])
    assert not np.array_equal(bits(g["read_cx"]), bits(g["made_cx"]))


def test_reader_restated_leaves_missing_voxels_free(tmp_path):
    """The reference keeps going when fscanf runs dry (:338-343): missing voxels stay free.  (The drop-in header
    reports a truncated file as an error instead; tests/test_gpu_gridfile.py.)"""
    g, _ = load()
    text = g["file_text"].tobytes()
    cut = text[:len(text) // 2]
    cut = cut[:cut.rfind(b" ") + 1]
    f = str(tmp_path / "t.in")
    open(f, "wb").write(cut)
    og, got = O.read_grid_in(f)
    n = int(np.prod(g["read_dims"][:3]))
    full = np.unpackbits(g["read_free_packed"])[:n]
    assert 0 < got < n and np.array_equal(og.free[:got], full[:got]) and np.all(og.free[got:] == 1)


import pytest


@pytest.mark.ref
@pytest.mark.skipif(not O.have_ref(), reason="compiled reference absent")
def test_live_reference_reader_on_whole_and_truncated_files(tmp_path):
    g, _ = load()
    text = g["file_text"].tobytes()
    cut = text[:len(text) // 3]
    cut = cut[:cut.rfind(b" ") + 1]
    for name, data in (("whole", text), ("cut", cut)):
        f = str(tmp_path / (name + ".in"))
        open(f, "wb").write(data)
        r = O.run_ref("voxelize", str(tmp_path / (name + ".waf")), gridin=f)
        og, got = O.read_grid_in(f)
        assert r["dims"].tolist() == [og.nx, og.ny, og.nz, og.wall]
        assert np.array_equal(r["free"], og.free), name
        assert np.array_equal(bits(r["cx"]), bits(og.cx)) and np.array_equal(bits(r["cy"]), bits(og.cy)) and np.array_equal(bits(r["cz"]), bits(og.cz))
