"""ASCII STL text made from triangle arrays: test inputs for the ASCII branch of STLReader (read_STL.hpp:99-129).  Used by
tests/golden/make_golden.py (which feeds the files to the reference's own reader) and by the tests (which rebuild the same bytes)."""
import numpy as np


def ascii_stl_text(tris, name="cubic", eol="\n", indent="  ", fmt="%.9g"):
    """An ASCII STL (the text form of the format: solid / facet normal / outer loop / vertex x 3 / endloop / endfacet / endsolid)
    of an (n, 12) triangle array; %.9g round-trips a float exactly."""
    out = ["solid " + name if name is not None else "solid"]
    for t in np.asarray(tris).reshape(-1, 12):
        out.append(indent + "facet normal " + " ".join(fmt % v for v in t[:3]))
        out.append(indent * 2 + "outer loop")
        for k in range(3):
            out.append(indent * 3 + "vertex " + " ".join(fmt % v for v in t[3 + 3 * k:6 + 3 * k]))
        out.append(indent * 2 + "endloop")
        out.append(indent + "endfacet")
    out.append("endsolid " + name if name is not None else "endsolid")
    return (eol.join(out) + eol).encode()


def ascii_stl_variants(tris, binary_file):
    """(tag, file bytes): well-formed ASCII STL files and the malformed ones that show how the reference's reader
    (read_STL.hpp:99-129) behaves -- every one deterministic (no read runs past the end of the file: each ends in a
    line the reader's loop stops at, or in a NUL byte)."""
    tris = np.asarray(tris, np.float32).reshape(-1, 12)
    std = ascii_stl_text(tris)
    v = [("standard", std),
         ("crlf_tabs_E", ascii_stl_text(tris, eol="\r\n", indent="\t", fmt="%.8E")),
         ("nameless_solid", ascii_stl_text(tris, name=None)),                    # `ss >> name >> name` swallows the first "facet": no triangle
         ("name_with_blank", std.replace(b"solid cubic", b"solid my part", 1)),  # the word after the name is not "facet": no triangle
         ("no_final_newline", std.rstrip(b"\n")),
         ("extra_blank_lines", std.replace(b"endfacet\n", b"endfacet\n\n   \n", 3))]   # getline x 3 then eats the wrong lines
    odd = std.split(b"\n")
    # number spellings num_get accepts / refuses: leading zeros, bare point, explicit plus, exponent forms
    odd[3] = b"      vertex +.5 -0 007.250"
    odd[4] = b"      vertex 1e3 1.e2 .5e-1"
    odd[5] = b"      vertex 1E+2 -1.5e-3 3."
    v.append(("number_spellings", b"\n".join(odd)))
    bad = std.split(b"\n")
    bad[3 + 7] = b"      vertex 0.25 abc 0.75"             # second triangle, first vertex: the stream fails in the middle of a vertex
    v.append(("garbage_coordinate", b"\n".join(bad)))
    nan = std.split(b"\n")
    nan[4 + 7] = b"      vertex nan 1 2"                   # libstdc++ does not read "nan"
    v.append(("nan_coordinate", b"\n".join(nan)))
    big = std.split(b"\n")
    big[5 + 7] = b"      vertex 1e39 -1e39 1e-50"          # overflow: +-FLT_MAX and failbit; underflow: 0, no failure
    v.append(("overflow_coordinate", b"\n".join(big)))
    dang = std.split(b"\n")
    dang[3 + 14] = b"      vertex 1e 2 3"                  # a dangling exponent
    v.append(("dangling_exponent", b"\n".join(dang)))
    lines = std.split(b"\n")
    v.append(("truncated_in_third_triangle", b"\n".join(lines[:1 + 14 + 5]) + b"\n\0"))   # ends behind the second vertex of triangle 3
    v.append(("truncated_behind_endfacet", b"\n".join(lines[:1 + 14]) + b"\0"))
    binary = bytearray(binary_file)
    binary[:80] = b"x" * 80                               # a BINARY file whose header is full to byte 79: sniffed as ASCII (:65)
    v.append(("binary_with_full_header", bytes(binary)))
    return v

