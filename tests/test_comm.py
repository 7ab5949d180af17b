"""The multi-GPU exchange behind the C ABI (include/weldacs.h wa_comm_*, csrc/host_comm.inc): an RCCL all-reduce (ncclMin)
of the per-generation best path cost (ACSRank_3D.hpp:263-264) on the communicator's own stream.  A 1-GPU box can only
form a world of ONE rank -- but with one rank the collective still runs through ncclCommInitRank / ncclAllReduce, which
is what these tests execute (no torch anywhere).  N > 1 on hardware is unmeasured (no multi-GPU node)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from welding_robot_amd import _lib, api, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/weldacs_multistart_%d" % os.getuid()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def compile_multistart():
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multistart_rccl.cpp"),
           "-L" + libdir, "-lweldacs", "-lpthread", "-Wl,-rpath," + libdir, "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr
    return EXE


def test_library_links_rccl_itself_and_the_cpp_host_compiles_without_torch():
    r = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True)
    assert "librccl" in r.stdout and "torch" not in r.stdout and "oracle" not in r.stdout
    compile_multistart()
    lib = _lib.load()
    import ctypes as C
    h = C.c_void_p()
    if lib.wa_ctx_create(0, C.byref(h)) == 0:
        lib.wa_ctx_destroy(h)
        pytest.skip("a HIP device is present")
    r = subprocess.run([EXE, "16", "8", "4", "all", "/tmp/weldacs_ms_cpu.txt"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_allreduce_best_with_one_rank_really_reduces_and_overlaps():
    """two searches in flight on one solver: global_best[g] = MIN over the active slots (k_min_over_slots) then
    ncclAllReduce(ncclMin) over the world of one rank; chunks are enqueued between wa_acs_run calls without waiting."""
    ctx = api.Context(0)
    og = O.synth_grid(40, seed=77, occ_prob=0.12)
    free = np.nonzero(og.free)[0]
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=2, max_colony=32)
    comm = api.Comm(ctx, 0, 1, api.Comm.unique_id())
    assert (comm.rank, comm.world) == (0, 1)
    K = 90
    p = api.default_params(max_iteration=K, predict=120.0, fixed_colony=32, rng_mode=api.RNG_DEV, seed=9)
    s.init_pheromone(1.0)
    s.begin(p, [int(free[0]), int(free[3])], [int(free[-1]), int(free[-7])], streams=[4, 11])
    with pytest.raises(api.WeldacsError):          # generations that were never enqueued cannot be exchanged
        comm.allreduce_best(s, 0, 10)
    done = 0
    for c in (25, 25, 40):
        s.run(c)
        comm.allreduce_best(s, done, c)
        done += c
    s.sync()
    glob = comm.read_best(0, K)
    t0, t1 = s.trace(0)["bestL"], s.trace(1)["bestL"]
    assert np.array_equal(bits(glob), bits(np.minimum(t0, t1)))
    assert np.isfinite(glob[-1]) and not np.array_equal(bits(t0), bits(t1))
    # the bookkeeping reductions a C++ launcher uses
    assert comm.allreduce([3.5, -2.0], "max").tolist() == [3.5, -2.0] and comm.allreduce([7.0], "sum").tolist() == [7.0]
    comm.barrier()
    with pytest.raises(api.WeldacsError):
        comm.read_best(0, 100000)
    comm.close(); s.close(); dg.close(); ctx.close()


@pytest.mark.gpu
def test_cpp_multistart_host_equals_the_oracle_trace():
    """examples/multistart_rccl.cpp (C4 from a C++ host: thread + wa_ctx + wa_comm per device) on the devices this box has:
    every rank's local history equals the oracle's DEV-mode run of that rank's problem, the global history is their MIN."""
    compile_multistart()
    n, ants, K = 48, 64, 60
    out = "/tmp/weldacs_ms_%d.txt" % os.getpid()
    r = subprocess.run([EXE, str(n), str(ants), str(K), "all", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    loc, glob = {}, {}
    for line in open(out):
        t = line.split()
        if t[0] == "local":
            loc.setdefault(int(t[1]), {})[int(t[2])] = int(t[3], 16)
        elif t[0] == "global":
            glob[int(t[1])] = int(t[2], 16)
    W = len(loc)
    assert W >= 1 and len(glob) == K
    hist = []
    for rk in range(W):
        og = O.synth_grid(n, seed=2024 + rk, occ_prob=0.10)
        free, *_ = synth.synth_grid(n, seed=2024 + rk, occ_prob=0.10)
        assert np.array_equal(free, og.free)
        sid, eid = og.resolve(np.zeros(3, np.float32)), og.resolve(np.full(3, n - 1, np.float32))
        a = O.Acs(og)
        tr = a.solve(sid, eid, K, float(np.float32(ants / 0.35)), fixed_colony=ants, mode=O.DEV, seed=12345 + rk, stream=rk)
        assert [loc[rk][g] for g in range(K)] == bits(tr["bestL"]).tolist(), rk
        hist.append(tr["bestL"])
    assert [glob[g] for g in range(K)] == bits(np.min(np.stack(hist), 0)).tolist()
