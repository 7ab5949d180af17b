#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref/ref_harness, built by
`make -C oracle`).  Every fixture = inputs + the reference's outputs for them; nothing here
is reference source.  The reference ships no tests and no golden files of its own,
so these vectors are the pin for oracle/weld_oracle.c (tests/test_oracle_golden.py) and,
through it, for the HIP path.

    python3 tests/golden/make_golden.py            # regenerate everything (~2 min)

Also copies the two binary STL *data* files the configs name (cubic.stl 684 B,
simplified_piece.stl 299 KB; `origin` mode: origin_piece.stl 1.5 MB) so that the GPU box -- which has no /root/reference -- can run
BASELINE configs C1/C2.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import waf  # noqa: E402
from stl_text import ascii_stl_text, ascii_stl_variants  # noqa: E402,F401

REFROOT = "/root/reference"
TMP = "/tmp/weld_golden"
os.makedirs(TMP, exist_ok=True)


def keep(d, keys):
    return {k: d[k] for k in keys if k in d}


def pack_free(d):
    d = dict(d)
    d["free_packed"] = np.packbits(d.pop("free"))
    return d


ACS_KEYS = ["dims", "precision", "start_id", "end_id", "points_ok", "best_L", "best_path", "best_choice",
            "colony_last", "lambda_last", "Q_last", "rand_calls", "next_rand", "pher_sum", "pher_hash",
            "tr_colony", "tr_finite", "tr_steps", "tr_bestL", "tr_iterbestL", "tr_lambda", "tr_Q"]


BSPLINE_CASES = [  # (tag, deg, ci, cf, n, tf, fill, pad, t0, dt, count)
    ("d0_main1", 0, 0, 0, 62, "150", "0", 0, "10", "10", 16),            # main.cpp:299-300 shape
    ("d2_main2", 2, 2, 2, 16, "6000", "0", 6, "50", "50", 121),          # main.cpp:337-338 shape
    ("d2_main2_fill1", 2, 2, 2, 16, "6000", "3f800000", 6, "50", "50", 121),
    ("d2_main2_fillnan", 2, 2, 2, 5, "77.7", "7fc00000", 0, "-3", "1.5", 60),
    ("d1", 1, 0, 0, 9, "1", "0", 0, "0", "0.03125", 40),
    ("d1c", 1, 1, 1, 33, "12.5", "bf000000", 0, "-1", "0.25", 60),
    ("d2", 2, 1, 1, 40, "2.5", "0", 0, "0", "0.05", 55),
    ("d3", 3, 2, 2, 25, "150", "0", 0, "-3", "3", 60),                   # DEGREE = CL+1: the well-defined use
    ("d3_asym", 3, 1, 2, 7, "9", "3f800000", 0, "0", "0.2", 50),
    ("d3_full", 3, 3, 3, 300, "6000", "0", 0, "0", "13.7", 450),
    ("d4", 4, 3, 3, 12, "33", "0", 0, "0", "0.5", 70),
    ("d5", 5, 2, 2, 1, "1", "0", 0, "0", "0.02", 55),
    ("d5_full", 5, 4, 4, 50, "100", "0", 0, "-2", "1", 110),
]


def gen_bspline():
    out = {}
    for tag, deg, ci, cf, n, tf, fill, pad, t0, dt, count in BSPLINE_CASES:
        g = O.run_ref("bspline", TMP + "/bs.waf", deg=deg, ci=ci, cf=cf, n=n, seed=1000 + n, tf=tf, fill=fill,
                      pad=pad, t0=t0, dt=dt, count=count)
        for k, v in g.items():
            out["%s/%s" % (tag, k)] = v
    waf.save(HERE + "/bspline_cases.waf", out)
    print("bspline cases", len(BSPLINE_CASES), os.path.getsize(HERE + "/bspline_cases.waf"), "bytes")


def gen_smooth():
    """main.cpp:273-352 end to end on cubic: pairs -> GTSP -> stitched path -> two smoothing passes
    (fixed sample times instead of clock()).  Same seed/arguments as pairs_cubic.waf."""
    pts = HERE + "/cubic_weld_points.in"
    for fill in ("0", "3f800000"):
        pr = O.run_ref("pairs", TMP + "/ps.waf", stl=HERE + "/cubic.stl", p="0.0219", wall=8, pts=pts, predict="0.5",
                       seed=4321, graph=TMP + "/graph_s.in", gtsp=1, smooth=1, fill=fill)
        keys = ["g_path_x", "g_path_y", "g_path_z", "tour_edges", "segments", "pair_len", "pair_paths_upper",
                "smooth_fill_bits"] + [k for k in pr if k.startswith("s1_") or k.startswith("s2_")]
        waf.save(HERE + "/smooth_cubic_fill%s.waf" % fill, keep(pr, keys))
        print("smooth", fill, len(pr["g_path_x"]), pr["s2_samples"].reshape(-1, 3)[[0, 60, -1]])


def gen_synth128(only=()):
    """synthetic 128^3 grid of BASELINE config C3 (our own PRNG), entering the reference through readGridMap"""
    g = O.synth_grid(128, seed=2024, occ_prob=0.10)
    gin = TMP + "/synth128.in"
    O.write_grid_in(g, gin)
    for tag, kw in [("acs_synth128_adaptive10", dict(gridin=gin, spt="0,0,0", ept="127,127,127", seed=12345, iters=10, predict="731.43", driven=1)),
                    ("acs_synth128_fixed256_4", dict(gridin=gin, spt="0,0,0", ept="127,127,127", seed=12345, iters=4, predict="731.43", fixed=256)),
                    ("acs_synth128_fixed256_40", dict(gridin=gin, spt="0,0,0", ept="127,127,127", seed=12345, iters=40, predict="731.43", fixed=256)),
                    # BASELINE config 3 at its stated 500 iterations (~100 s of the reference on one core): the field crosses the
                    # denormal range and reaches the 2-ulp fixed point
                    ("acs_synth128_fixed256_500", dict(gridin=gin, spt="0,0,0", ept="127,127,127", seed=12345, iters=500, predict="731.43", fixed=256))]:
        if only and tag not in only:
            continue
        a = O.run_ref("acs", TMP + "/a.waf", **kw)
        out = keep(a, ACS_KEYS)
        out["args"] = np.frombuffer(repr(sorted((k, str(v)) for k, v in kw.items() if k != "gridin")).encode(), np.uint8)
        out["grid_seed"] = np.array([2024], np.int64)
        out["grid_free_count"] = np.array([int(g.free.sum())], np.int64)
        out["grid_fnv"] = np.array([O.fnv1a_bytes(g.free.tobytes()) - (1 << 64) if O.fnv1a_bytes(g.free.tobytes()) >= (1 << 63) else O.fnv1a_bytes(g.free.tobytes())], np.int64)
        waf.save(HERE + "/%s.waf" % tag, out)
        print(tag, a["best_L"], a["tr_colony"], a["tr_steps"])



def gen_gridfile():
    """The grid-map text cache: the file the reference WRITES for cubic.stl
    (model_grid_map.hpp:275-294, with the last-triangle bounding box and "%f" of Q5) and what the reference's own
    readGridMap (:300-356) rebuilds from that file -- which is not the grid that was written."""
    cubic = os.path.join(HERE, "cubic.stl")
    gf = TMP + "/cubic_grid_ref.in"
    made = O.run_ref("voxelize", TMP + "/vg.waf", stl=cubic, p="0.0219", wall=8, gridout=gf)
    back = O.run_ref("voxelize", TMP + "/vr.waf", gridin=gf)
    out = {"file_text": np.frombuffer(open(gf, "rb").read(), np.uint8).copy(),
           "made_dims": made["dims"], "made_precision": made["precision"], "made_cx": made["cx"], "made_cy": made["cy"],
           "made_cz": made["cz"], "made_free_packed": np.packbits(made["free"]),
           "read_dims": back["dims"], "read_precision": back["precision"], "read_cx": back["cx"], "read_cy": back["cy"],
           "read_cz": back["cz"], "read_free_packed": np.packbits(back["free"])}
    waf.save(HERE + "/gridfile_cubic.waf", out)
    print("gridfile_cubic.waf: %d bytes of file text, dims %s" % (out["file_text"].size, made["dims"][:3].tolist()))


def gen_nb26():
    """reference member functions on 26-neighbour adjacency (harness nb=26, always driven)."""
    cubic = os.path.join(HERE, "cubic.stl")
    piece = os.path.join(HERE, "simplified_piece.stl")
    cases = [
        ("acs_cubic_nb26_adaptive", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=50, predict="1.03", nb=26)),
        ("acs_cubic_nb26_fixed16", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=777, iters=50, predict="1.03", fixed=16, nb=26)),
        ("acs_cubic_nb26_seam", dict(stl=cubic, p="0.0225", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=10, predict="1.03", nb=26)),
        ("acs_piece_nb26_fixed128", dict(stl=piece, p="0.0148", wall=4, snode="0,0,0", enode="22,32,63", seed=12345, iters=100, predict="5.4126", fixed=128, nb=26)),
    ]
    for tag, kw in cases:
        a = O.run_ref("acs", TMP + "/a.waf", **kw)
        out = keep(a, ACS_KEYS)
        out["args"] = np.frombuffer(repr(sorted((k, str(v) if k != "stl" else os.path.basename(v)) for k, v in kw.items())).encode(), np.uint8)
        waf.save(HERE + "/%s.waf" % tag, out)
        print(tag, a["best_L"], len(a["best_path"]), a["tr_steps"][:4])
    g = O.synth_grid(64, seed=77, occ_prob=0.10)
    gin = TMP + "/synth64.in"
    O.write_grid_in(g, gin)
    kw = dict(gridin=gin, spt="0,0,0", ept="63,63,63", seed=4242, iters=6, predict="200", fixed=64, nb=26)
    a = O.run_ref("acs", TMP + "/a.waf", **kw)
    out = keep(a, ACS_KEYS)
    out["args"] = np.frombuffer(repr(sorted((k, str(v)) for k, v in kw.items() if k != "gridin")).encode(), np.uint8)
    h = O.fnv1a_bytes(g.free.tobytes())
    out["grid_fnv"] = np.array([h - (1 << 64) if h >= (1 << 63) else h], np.int64)
    waf.save(HERE + "/acs_synth64_nb26_fixed64.waf", out)
    print("synth64 nb26", a["best_L"], a["tr_steps"], a["tr_finite"])


def gen_ascii():
    """SURVEY 8(a) a1, the ASCII branch of STLReader (read_STL.hpp:99-129): files made HERE from cubic.stl's twelve triangles,
    read by the reference's own reader (harness command `stl`), plus the reference's voxelisation of the standard one (Q11: normals
    stay 0, so every voxel of a triangle's bounding box +- p is occupied)."""
    cubic = O.stl_parse(open(os.path.join(HERE, "cubic.stl"), "rb").read())
    out = {}
    tags = []
    for tag, data in ascii_stl_variants(cubic, open(os.path.join(HERE, "cubic.stl"), "rb").read()):
        f = TMP + "/ascii_%s.stl" % tag
        open(f, "wb").write(data)
        r = O.run_ref("stl", TMP + "/s.waf", stl=f)
        out["file_" + tag] = np.frombuffer(data, np.uint8).copy()
        out["n_" + tag] = r["n_tris"]
        out["tris_" + tag] = np.asarray(r["tris"], np.float32)
        tags.append(tag)
        print("%-30s %5d bytes -> %3d triangles" % (tag, len(data), int(np.asarray(r["n_tris"]).reshape(-1)[0])))
    out["tags"] = np.frombuffer(" ".join(tags).encode(), np.uint8).copy()
    waf.save(HERE + "/stl_ascii.waf", out)
    f = os.path.join(HERE, "cubic_ascii.stl")
    open(f, "wb").write(ascii_stl_text(cubic))
    v = O.run_ref("voxelize", TMP + "/va.waf", stl=f, p="0.0219", wall=8)
    v = pack_free(v)
    v.pop("t_voxelize", None)
    waf.save(HERE + "/vox_cubic_ascii_p0219_w8.waf", v)
    print("vox_cubic_ascii_p0219_w8: dims %s, %d free" % (v["dims"][:3].tolist(), int(np.unpackbits(v["free_packed"]).sum())))
    # the work piece as text (5 977 triangles that are NOT axis-aligned: with the normals gone the occupancy differs from the binary file's)
    piece = O.stl_parse(open(os.path.join(HERE, "simplified_piece.stl"), "rb").read())
    f = TMP + "/piece_ascii.stl"
    open(f, "wb").write(ascii_stl_text(piece, name="piece"))
    v = O.run_ref("voxelize", TMP + "/vp.waf", stl=f, p="0.0148", wall=4)
    v = pack_free(v)
    v["tris_head"] = v["tris"][:12 * 16].copy()
    v["tris_sum"] = np.array([np.sum(v["tris"].astype(np.float64))])
    del v["tris"]
    v.pop("t_voxelize", None)
    waf.save(HERE + "/vox_piece_ascii_p0148_w4.waf", v)
    b = waf.load(HERE + "/vox_piece_p0148_w4.waf")
    print("vox_piece_ascii_p0148_w4: dims %s, %d free (the binary file: %d free)" % (v["dims"][:3].tolist(), int(np.unpackbits(v["free_packed"]).sum()),
                                                                                  int(np.unpackbits(b["free_packed"]).sum())))


def gen_origin():
    """The reference's largest mesh (files/origin_piece.stl, 29 888 triangles: the un-simplified work piece SURVEY 8(f) N1 quotes) voxelised
    by the reference's creatGridMap at precision 0.0100, wall 4 (91 x 45 x 30 voxels: ~30 s of the reference's O(T N^3) loop)."""
    f = "origin_piece.stl"
    shutil.copyfile(os.path.join(REFROOT, "files", f), os.path.join(HERE, f))
    os.chmod(os.path.join(HERE, f), 0o644)
    v = O.run_ref("voxelize", TMP + "/vo.waf", stl=os.path.join(HERE, f), p="0.0100", wall=4)
    v = pack_free(v)
    v["tris_head"] = v["tris"][:12 * 16].copy()
    v["tris_sum"] = np.array([np.sum(v["tris"].astype(np.float64))])
    del v["tris"]
    print("reference voxelisation: %.2f s" % float(np.asarray(v.pop("t_voxelize", [0.0])).reshape(-1)[0]))
    waf.save(HERE + "/vox_origin_p0100_w4.waf", v)
    # ... and the reference's search across that grid, corner to corner (mesh -> grid -> ACS_Rank on the third data file)
    kw = dict(stl=os.path.join(HERE, f), p="0.0100", wall=4, snode="0,0,0", enode="29,44,90", seed=2468, iters=100, predict="2.0", fixed=64)
    a = O.run_ref("acs", TMP + "/ao.waf", **kw)
    out = keep(a, ACS_KEYS)
    out["args"] = np.frombuffer(repr(sorted((k, str(v) if k != "stl" else os.path.basename(v)) for k, v in kw.items())).encode(), np.uint8)
    waf.save(HERE + "/acs_origin_fixed64.waf", out)
    print("acs_origin_fixed64", a["best_L"], len(a["best_path"]))


def main():
    assert O.have_ref(), "build oracle/_ref first: make -C oracle"
    if len(sys.argv) > 1 and sys.argv[1] == "bspline":      # regenerate only the trajectory fixtures
        gen_bspline()
        gen_smooth()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nb26":
        gen_nb26()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "synth128":
        gen_synth128(only=sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gridfile":
        gen_gridfile()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "origin":
        gen_origin()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ascii":
        gen_ascii()
        return
    for f in ("cubic.stl", "simplified_piece.stl"):
        shutil.copyfile(os.path.join(REFROOT, "files", f), os.path.join(HERE, f))
        os.chmod(os.path.join(HERE, f), 0o644)
    cubic = os.path.join(HERE, "cubic.stl")
    piece = os.path.join(HERE, "simplified_piece.stl")

    # ---- libc / libstdc++ known answers ------------------------------------------------
    r = O.run_ref("rand", TMP + "/rand.waf", seed=12345, n=64)
    waf.save(HERE + "/rand_12345.waf", r)
    sorts = {}
    for i, (n, lv, sd) in enumerate([(16, 3, 1), (17, 2, 2), (64, 4, 3), (256, 8, 4), (257, 1, 5), (1000, 50, 6)]):
        s = O.run_ref("sort", TMP + "/sort.waf", n=n, levels=lv, seed=sd)
        sorts["keys%d" % i] = s["keys"]
        sorts["perm%d" % i] = s["perm"]
    waf.save(HERE + "/std_sort.waf", sorts)

    # ---- voxelisation (KA1) ---------------------------------------------------------------
    for tag, stl, p, wall in [("cubic_p0219_w8", cubic, "0.0219", 8), ("cubic_p0225_w8", cubic, "0.0225", 8),
                              ("piece_p0148_w4", piece, "0.0148", 4)]:
        v = O.run_ref("voxelize", TMP + "/v.waf", stl=stl, p=p, wall=wall)
        v = pack_free(v)
        if v["tris"].size > 12 * 16:
            v["tris_head"] = v["tris"][:12 * 16].copy()
            v["tris_sum"] = np.array([np.sum(v["tris"].astype(np.float64))])
            del v["tris"]
        v.pop("t_voxelize", None)
        waf.save(HERE + "/vox_%s.waf" % tag, v)

    # ---- ACS_Rank (KA2 and friends) -------------------------------------------------------
    cases = [
        ("acs_cubic_ka2_native", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=50, predict="1.03")),
        ("acs_cubic_ka2_driven", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=50, predict="1.03", driven=1)),
        ("acs_cubic_predict5", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=150, predict="5", driven=1)),
        ("acs_cubic_fixed16", dict(stl=cubic, p="0.0219", wall=8, snode="4,4,4", enode="20,27,20", seed=777, iters=50, predict="1.03", fixed=16)),
        ("acs_cubic_seam", dict(stl=cubic, p="0.0225", wall=8, snode="4,4,4", enode="20,27,20", seed=12345, iters=10, predict="1.03", driven=1)),
        ("acs_piece_adaptive", dict(stl=piece, p="0.0148", wall=4, snode="0,0,0", enode="22,32,63", seed=12345, iters=200, predict="5.4126", driven=1)),
        ("acs_piece_fixed128", dict(stl=piece, p="0.0148", wall=4, snode="0,0,0", enode="22,32,63", seed=12345, iters=200, predict="5.4126", fixed=128)),
    ]
    for tag, kw in cases:
        a = O.run_ref("acs", TMP + "/a.waf", **kw)
        out = keep(a, ACS_KEYS)
        out["args"] = np.frombuffer(repr(sorted((k, str(v) if k != "stl" else os.path.basename(v)) for k, v in kw.items())).encode(), np.uint8)
        waf.save(HERE + "/%s.waf" % tag, out)
        print(tag, a["best_L"], len(a["best_path"]))

    gen_synth128()

    # ---- pair flow + GTSP on cubic (main.cpp:279-283) --------------------------------------------
    vg = O.run_ref("voxelize", TMP + "/v.waf", stl=cubic, p="0.0219", wall=8)
    cx, cy, cz = vg["cx"], vg["cy"], vg["cz"]
    nodes = [(4, 4, 4), (20, 27, 20), (4, 27, 20), (20, 4, 4), (12, 2, 12)]
    pts = TMP + "/cubic_weld_points.in"
    with open(pts, "w") as f:
        f.write("%d\n" % len(nodes))
        for z, y, x in nodes:
            f.write("%f %f %f\n" % (cx[x], cy[y], cz[z]))
    shutil.copyfile(pts, HERE + "/cubic_weld_points.in")
    pr = O.run_ref("pairs", TMP + "/p.waf", stl=cubic, p="0.0219", wall=8, pts=pts, predict="0.5", seed=4321,
                   graph=TMP + "/graph.in", gtsp=1)
    waf.save(HERE + "/pairs_cubic.waf", pr)
    print("pairs", pr["pair_cost"].reshape(5, 5), waf.text(pr, "graph_text")[:40].encode(), pr["tour_edges"], pr["tour_L"], pr["gtsp_iters"])

    # ---- GTSP alone (KA4) ----------------------------------------------------------------------
    def write_graph(path, n, fn):
        with open(path, "w") as f:
            f.write("%d %d\n" % (n, n * (n - 1) // 2))
            for i in range(n):
                for j in range(i + 1, n):
                    f.write("%s\n" % fn(i, j))
    write_graph(TMP + "/g8.in", 8, lambda i, j: "%.1f" % (((7 * i + 13 * j) % 10 + 1) / 10.0))
    g8 = O.run_ref("gtsp", TMP + "/g8.waf", graph=TMP + "/g8.in", seed=1)
    g8.pop("t_gtsp", None)
    waf.save(HERE + "/gtsp_ka4_n8.waf", g8)
    print("gtsp8", g8["tour_L"], g8["gtsp_iters"], g8["tour_edges"][::2] + 1)
    st = np.uint64(4242)
    rs = np.random.RandomState(4242)
    P = rs.randint(0, 100, size=(64, 3))
    write_graph(TMP + "/g64.in", 64, lambda i, j: "%.3f" % (np.abs(P[i] - P[j]).sum() / 100.0 + 0.001))
    g64 = O.run_ref("gtsp", TMP + "/g64.waf", graph=TMP + "/g64.in", seed=4242)
    g64.pop("t_gtsp", None)
    waf.save(HERE + "/gtsp_n64.waf", g64)
    print("gtsp64", g64["tour_L"], g64["gtsp_iters"])

    # ---- BS_Basic trajectory smoothing ------------------------------------------
    gen_bspline()
    gen_smooth()
    gen_nb26()
    gen_gridfile()


if __name__ == "__main__":
    main()
