"""Build libweldacs.so (the HIP/gfx950 product library) in-tree with hipcc.

    python -m welding_robot_amd.build            # rebuild if sources changed

Flags that matter for parity with the reference's fp32 semantics:
  -ffp-contract=off                      no FMA contraction (x86-64 g++ without -mfma never contracts)
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE '/' and sqrtf (hipcc default, spelled out)
  -fno-fast-math, denormals kept (no -fgpu-flush-denormals-to-zero)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libweldacs.so")
# the same library with the forced-hand-back knobs of tests/test_gpu_reentry.py compiled in (-DWA_TEST_KNOBS); never the product
KNOBS_LIB_PATH = os.path.join(LIB_DIR, "libweldacs_knobs.so")
# the knobs library with ONE regression put back on purpose (-DWA_BEST_READ_LATE: k_evap_rank_mark's publishing block reads the old best behind its barriers):
# the negative half of tests/test_gpu_late_waves.py -- the knob must be able to open the window it guards.  Never the product
LATE_READ_LIB_PATH = os.path.join(LIB_DIR, "libweldacs_knobs_late_read.so")
SOURCES = ["weldacs.hip"]
DEPS = ["wa_device.h", "acs_kernels.hpp", "acs_dev.hpp", "acs_walk.hpp", "acs_update.hpp", "acs_nb26.hpp", "walk_loop_gfx950.hpp", "grid_kernels.hpp", "gtsp_kernels.hpp", "traj_kernels.hpp",
        "stl_text.hpp", "host_grid.inc", "host_acs.inc", "host_gtsp.inc", "host_traj.inc", "host_comm.inc"]
HEADER = os.path.join(os.path.dirname(HERE), "include", "weldacs.h")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-Wno-inline-asm", "-Wno-pass-failed"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def rocm_lib_dir():
    """where librccl / libamdhip64 of the chosen toolchain live: $ROCM_PATH/lib, else next to the hipcc in use"""
    import shutil
    root = os.environ.get("ROCM_PATH")
    if not root:
        exe = shutil.which(hipcc()) or hipcc()
        root = os.path.dirname(os.path.dirname(os.path.realpath(exe)))
    d = os.path.join(root, "lib")
    return d if os.path.isdir(d) else "/opt/rocm/lib"


def needs_build(lib=None):
    lib = lib or LIB_PATH
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    files = [os.path.join(CSRC, f) for f in SOURCES + DEPS] + [HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=False, extra=(), out=None):
    """out: alternative output path (experiment variants built with extra -D flags)"""
    if out is None and not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    out = out or LIB_PATH
    # librccl: the global-best all-reduce of the multi-GPU path (csrc/host_comm.inc) -- linked by the library itself
    # (found through the toolchain's own directory, which also becomes the library's run path: a ROCm outside /opt/rocm loads)
    libdir = rocm_lib_dir()
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out, "-L" + libdir, "-Wl,-rpath," + libdir, "-lrccl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if os.environ.get("WA_BUILD_CHECK", "1") != "0":
        check_isa(extra)
    return out


def check_isa(extra=()):
    """What the 26-neighbour fast loop relies on beyond the compiler's promises (loads issued by one inline statement and waited for by a
    later one, touch loads in v250..v253) is verified on the assembled device code of THIS toolchain after every build
    (tools/check_walk26_isa.py; WA_BUILD_CHECK=0 skips): a compiler that puts a copy or a spill between issue and wait fails the
    build with a message instead of producing wrong walks."""
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_walk26_isa.py")
    if not os.path.exists(tool) or any(str(e).startswith("-DWA_ASM_SPAN") or str(e) in ("-DWA_STAMPS", "-DWA_ASM_STAMPS") for e in extra):
        return   # (an installed package without the tools directory; diagnostic builds that restructure the loops)
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libweldacs: the assembled k_walk_dev26 does not keep what its inline statements rely on with this toolchain (%s):\n%s%s"
                           % (hipcc(), r.stdout, r.stderr[-2000:]))


def build_knobs(force=False, verbose=False):
    """lib/libweldacs_knobs.so: what tests/test_gpu_reentry.py loads (api.Context(lib_path=...))"""
    if not force and not needs_build(KNOBS_LIB_PATH):
        return KNOBS_LIB_PATH
    return build(force=True, verbose=verbose, extra=["-DWA_TEST_KNOBS"], out=KNOBS_LIB_PATH)


def build_late_read(force=False, verbose=False):
    """lib/libweldacs_knobs_late_read.so (see LATE_READ_LIB_PATH)"""
    if not force and not needs_build(LATE_READ_LIB_PATH):
        return LATE_READ_LIB_PATH
    return build(force=True, verbose=verbose, extra=["-DWA_TEST_KNOBS", "-DWA_BEST_READ_LATE"], out=LATE_READ_LIB_PATH)


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    if "--knobs" in sys.argv:
        print(build_knobs(force="--force" in sys.argv, verbose=True))
        print(build_late_read(force="--force" in sys.argv, verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None,
                extra=defs + (["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])))
