"""CPU sanitizer leg (the reference has no sanitizer flags and several latent UB sites; the build's CPU side runs under
ASan + UBSan).  Nothing here touches a GPU: GPU sanitizers / XNACK are not available on the pool.

  * the oracle (oracle/weld_oracle.c, `make -C oracle asan`) over the golden vectors, in a child interpreter with libasan preloaded;
  * every host-side C++ file of the repo that is not compiled by hipcc -- the RCCL stand-in (tests/mock_rccl), the drop-in consumers
    (tests/cpp/*.cpp, examples/*.cpp) with the drop-in headers they include -- compiled with -fsanitize=address,undefined -Wall -Wextra,
    and run as far as a box without a HIP device lets them: up to the library's "no device" error, through the headers' failure paths
    (constructors, error returns, destructors of objects whose context never came up).
"""
import os
import subprocess
import sys

import pytest
from tmpw import TMPW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
OUT = TMPW + "weldacs_san_%d" % os.getuid()
G = os.path.join(ROOT, "tests", "golden")


def runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p):
        pytest.skip("%s not found next to gcc" % name)
    return p


def san_env(**extra):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.update(extra)
    return env


def clean_of_reports(text):
    return "ERROR: AddressSanitizer" not in text and "runtime error:" not in text and "ERROR: LeakSanitizer" not in text


def test_oracle_over_the_goldens_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "oracle", "_asan", "libweld_oracle_asan.so")
    # (the 500- and 40-generation 128^3 goldens are minutes under ASan: the other 53 cover every entry point of the oracle)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
           os.path.join(ROOT, "tests", "test_trajectory_golden.py"), os.path.join(ROOT, "tests", "test_gridfile.py"), "-k", "not fixed256_500 and not fixed256_40"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=san_env(LD_PRELOAD=runtime("libasan.so"), WELD_ORACLE_LIB=lib))
    out = r.stdout + r.stderr
    assert r.returncode == 0 and clean_of_reports(out), out[-4000:]
    assert " passed" in out and "failed" not in out
    # ... and the sanitizers really are in that process: an out-of-bounds read through the same library must be reported
    probe = ("import ctypes as C, oracle_lib as O; L = O.lib(); import numpy as np; k = np.arange(4096, dtype=np.float32); p = np.zeros(1024, np.int32); "
             "L.wo_std_sort_perm(k.ctypes.data_as(C.c_void_p), C.c_int32(4096), p.ctypes.data_as(C.c_void_p)); print('survived')")
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, cwd=os.path.join(ROOT, "tests"),
                       env=san_env(LD_PRELOAD=runtime("libasan.so"), WELD_ORACLE_LIB=lib))
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr, (r.returncode, r.stderr[-1500:])


def host_sources():
    from welding_robot_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "welding_robot_amd", "include")]
    link = ["-L" + libdir, "-lweldacs", "-lpthread", "-Wl,-rpath," + libdir]
    return [
        ("mock_rccl", os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), ["-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"], [], None),
        ("shard_check", os.path.join(ROOT, "tests", "cpp", "shard_check.cpp"), inc, link,
         [os.path.join(G, "cubic.stl"), "0.0219", "8", os.path.join(G, "cubic_weld_points.in"), "0.5", "99", "0,0", TMPW + "weldacs_san_shard.txt", "3"]),
        ("gridfile_check", os.path.join(ROOT, "tests", "cpp", "gridfile_check.cpp"), inc, link,
         ["write", os.path.join(G, "cubic.stl"), "0.0219", "8", TMPW + "weldacs_san_grid.in", "0", TMPW + "weldacs_san_grid.txt"]),
        ("dropin_demo", os.path.join(ROOT, "examples", "dropin_demo.cpp"), inc, link,
         [os.path.join(G, "cubic.stl"), "0.0219", "8", os.path.join(G, "cubic_weld_points.in"), "0.5", TMPW + "weldacs_san_graph.in", "dev", "3", TMPW + "weldacs_san_demo.txt"]),
        ("multistart_rccl", os.path.join(ROOT, "examples", "multistart_rccl.cpp"), inc, link, ["16", "8", "4", "all", TMPW + "weldacs_san_ms.txt"]),
        ("scalar_calls", os.path.join(ROOT, "examples", "scalar_calls.cpp"), inc, link, [os.path.join(G, "cubic.stl"), "0.0219", "8", TMPW + "weldacs_san_scalar.txt"]),
    ]


@pytest.mark.parametrize("name", ["mock_rccl", "shard_check", "gridfile_check", "dropin_demo", "multistart_rccl", "scalar_calls"])
def test_host_side_cpp_compiles_and_runs_clean_under_asan_and_ubsan(name):
    from welding_robot_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    os.makedirs(OUT, exist_ok=True)
    _, src, cflags, link, argv = next(h for h in host_sources() if h[0] == name)
    exe = os.path.join(OUT, name + (".so" if "-shared" in cflags else ""))
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra"] + SAN + cflags + [src] + link + ["-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "warning" not in r.stderr, r.stderr[-3000:]
    if argv is None:
        return   # (a shared object, or a program that needs files of its own: compiled and linked is what a GPU-less box can check)
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    if lib.wa_ctx_create(0, C.byref(h)) == 0:
        lib.wa_ctx_destroy(h)
        pytest.skip("a HIP device is present: the sanitizer leg is CPU-only")
    # the HIP runtime inside libweldacs.so is not instrumented: ASan only needs to come first in the executable's own link order (it does)
    r = subprocess.run([exe] + argv, capture_output=True, text=True, env=san_env(), timeout=120)
    out = r.stdout + r.stderr
    assert clean_of_reports(out), out[-3000:]
    # no device: the program says so and ends by itself (a crash would be a signal = a negative code; gridfile_check reports the
    # failure in its dump file and returns 0, the others return their error code)
    assert r.returncode >= 0 and "no CPU fallback" in out, (r.returncode, out[-1500:])


def test_ascii_stl_reader_under_asan_and_ubsan():
    """The library's ASCII STL reader (csrc/stl_text.hpp) is plain host C++: compiled on its own with the sanitizers and fed the committed text
    files, damaged ones and files cut at every kind of place -- exactly sized heap buffers without a terminator, so a read one byte past the
    text is reported -- and held against what libweldacs.so (the same code, built by hipcc) returns for the same bytes."""
    import numpy as np
    import stl_text
    import waf
    import oracle_lib as O
    from welding_robot_amd import api
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "stl_text_check")
    cmd = ["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror"] + SAN + ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "stl_text_check.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    g = waf.load(os.path.join(G, "stl_ascii.waf"))
    datas = [bytes(g["file_" + t]) for t in bytes(g["tags"]).decode().split()]
    rs = np.random.RandomState(11)
    text = stl_text.ascii_stl_text(O.stl_parse(open(os.path.join(G, "cubic.stl"), "rb").read()))
    junk = [b"facet", b"vertex", b"1e", b".", b"-", b"+.e1", b"inf", b"1e400", b"12abc", b"\n", b"\r\n", b"0" * 600, b"9" * 700 + b"e9999"]
    for case in range(150):
        toks = text.replace(b"\n", b" \n ").split(b" ")
        for _ in range(int(rs.randint(1, 6))):
            k = int(rs.randint(0, len(toks)))
            toks[k:k + int(rs.randint(0, 3))] = [junk[int(rs.randint(0, len(junk)))]]
        d = b" ".join(toks)
        datas.append(d[:int(rs.randint(80, len(d) + 1))] if case % 2 else d)
    datas.append(b"solid s\n" + b" " * 80 + b"facet")
    files = []
    for i, d in enumerate(datas):
        f = os.path.join(OUT, "stl_%03d.stl" % i)
        open(f, "wb").write(d)
        files.append(f)
    r = subprocess.run([exe] + files, capture_output=True, text=True, env=san_env())
    assert r.returncode == 0 and clean_of_reports(r.stdout + r.stderr), (r.stdout[-1500:] + r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(datas)
    refused = 0
    for d, line in zip(datas, lines):
        n, n2, n3, h = line.split()
        try:
            t = api.stl_parse(d)
        except api.WeldacsError as e:
            assert int(n) == -e.code == -5          # WA_ERR_FORMAT on both sides
            refused += 1
            continue
        assert int(n) == int(n2) == len(t)
        assert int(n3) == (-7 if len(t) else 0)     # WA_ERR_CAPACITY into a buffer one triangle short
        assert int(h, 16) == O.fnv1a_bytes(np.ascontiguousarray(t, np.float32).tobytes())
    assert refused >= 1
