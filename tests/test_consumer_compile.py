"""The reference's real consumer -- /root/reference/main.cpp, unmodified -- compiled against the drop-in headers
the way INTEGRATION.md recipe A / welding_robot_amd/cmake/weldacs_dropin.cmake prescribe (main.cpp:1-11,
:33-35, :268-352).  main.cpp is copied to a temp directory AT TEST TIME (never into the repo): quoted includes are
looked up next to the including file first, so only a copy outside the reference tree lets -I order decide.
Needs /root/reference (this container only), hence `ref`."""
import os
import shutil
import subprocess
import sysconfig
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FIVE = ["BSplineBasic.h", "ACSRank_3D.hpp", "read_STL.hpp", "ACS_GTSP.hpp", "model_grid_map.hpp"]
DROPIN = os.path.join(ROOT, "welding_robot_amd", "include", "core")

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.cpp")), reason="reference tree absent")]


def flags(dropin_first):
    import numpy
    inc = [REF, REF + "/common", REF + "/core", REF + "/coppeliaSim-client", REF + "/coppeliaSim-client/include",
           REF + "/coppeliaSim-client/include/stack", REF + "/coppeliaSim-client/remoteApi",
           sysconfig.get_paths()["include"], numpy.get_include()]
    if dropin_first:  # include_directories(BEFORE ...) of weldacs_dropin.cmake
        inc = [os.path.join(ROOT, "include"), os.path.join(ROOT, "welding_robot_amd", "include")] + inc
    return ["-std=c++14", "-DNON_MATLAB_PARSING", "-DMAX_EXT_API_CONNECTIONS=255", "-DDO_NOT_USE_SHARED_MEMORY"] + ["-I" + d for d in inc]


def resolved(stderr):
    """header basename -> set of directories `g++ -H` reports for it"""
    got = {}
    for line in stderr.splitlines():
        if line.startswith(".") and " " in line:
            path = line.split(" ", 1)[1].strip()
            if os.path.basename(path) in FIVE:
                got.setdefault(os.path.basename(path), set()).add(os.path.dirname(os.path.realpath(path)))
    return got


def test_main_cpp_shadow_copy_binds_the_dropin_headers_and_the_c_abi():
    tmp = tempfile.mkdtemp(prefix="weldacs_consumer_")
    try:
        shutil.copy(os.path.join(REF, "main.cpp"), os.path.join(tmp, "main.cpp"))
        obj = os.path.join(tmp, "main.o")
        r = subprocess.run(["g++"] + flags(True) + ["-H", "-c", "main.cpp", "-o", obj], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        got = resolved(r.stderr)
        assert set(got) == set(FIVE), got
        for h, dirs in got.items():
            assert dirs == {os.path.realpath(DROPIN)}, (h, dirs)   # none of the five comes from /root/reference/core
        # the consumer object now calls the C ABI (the drop-in classes are header-only wrappers over it)
        und = subprocess.run(["nm", "-u", "-C", obj], capture_output=True, text=True).stdout
        for sym in ("wa_ctx_create", "wa_stl_read_file", "wa_grid_from_mesh", "wa_grid_resolve_points", "wa_acs_solve",
                    "wa_gtsp_solve", "wa_bspline_create", "wa_bspline_eval"):
            assert sym in und, sym
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_in_place_compile_would_keep_the_reference_headers():
    """Negative control for the recipe: with main.cpp left where it is, the same -I order still resolves all five
    planning headers to the reference's own core/ -- which is why the recipe compiles a shadow copy."""
    r = subprocess.run(["g++"] + flags(True) + ["-H", "-fsyntax-only", os.path.join(REF, "main.cpp")], capture_output=True, text=True,
                       cwd=tempfile.gettempdir())
    assert r.returncode == 0, r.stderr[-3000:]
    got = resolved(r.stderr)
    for h in FIVE:
        assert got[h] == {os.path.realpath(REF + "/core")}, (h, got[h])


def test_cmake_fragment_builds_the_reference_application_on_the_c_abi():
    """INTEGRATION.md recipe A.1 run literally: a COPY of the reference tree (made at test time in a temp directory, never in the
    repo), the two lines `file(GLOB USER_SOURCE ...)` / `add_executable(...)` of its CMakeLists.txt replaced by
    `set(WELDACS ...)` + `include(weldacs_dropin.cmake)`, then cmake configure + build.  The resulting `welding_robot`
    executable -- the reference's whole application shell: menu, simulator client, plotting -- links libweldacs.so and
    its planning calls are the wa_* entry points."""
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    from welding_robot_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    tmp = tempfile.mkdtemp(prefix="weldacs_cmake_")
    try:
        src, bld = os.path.join(tmp, "src"), os.path.join(tmp, "build")
        shutil.copytree(REF, src)
        cml = os.path.join(src, "CMakeLists.txt")
        text = open(cml).read()
        old = 'file(GLOB USER_SOURCE "*.cpp")\nadd_executable(${PROJECT_NAME} ${USER_SOURCE})\n'
        assert old in text
        open(cml, "w").write(text.replace(old, "set(WELDACS %s)\ninclude(${WELDACS}/welding_robot_amd/cmake/weldacs_dropin.cmake)\n" % ROOT))
        os.makedirs(bld)
        r = subprocess.run(["cmake", src, "-DCMAKE_BUILD_TYPE=Release"], cwd=bld, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        r = subprocess.run(["cmake", "--build", ".", "-j4"], cwd=bld, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        exe = os.path.join(src, "build", "bin", "welding_robot")   # EXECUTABLE_OUTPUT_PATH of the reference's CMakeLists.txt
        assert os.path.exists(exe)
        assert "libweldacs.so" in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        und = subprocess.run(["nm", "-u", "-C", exe], capture_output=True, text=True).stdout
        for sym in ("wa_ctx_create", "wa_stl_read_file", "wa_grid_from_mesh", "wa_acs_solve", "wa_gtsp_solve", "wa_bspline_eval"):
            assert sym in und, sym
        # the shadow copy is what was compiled, the original main.cpp is untouched
        assert os.path.exists(os.path.join(bld, "weldacs_shadow", "main.cpp"))
        assert open(os.path.join(src, "main.cpp"), "rb").read() == open(os.path.join(REF, "main.cpp"), "rb").read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
