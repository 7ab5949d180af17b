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
from tmpw import TMPW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = TMPW + "weldacs_multistart_%d" % os.getuid()


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
    r = subprocess.run([EXE, "16", "8", "4", "all", TMPW + "weldacs_ms_cpu.txt"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout


def test_the_mock_stands_in_for_every_rccl_entry_point_the_library_calls():
    """tests/mock_rccl (what tests/test_gpu_mock_ranks.py puts in front of librccl to run world > 1 on one GPU) compiles here and exports
    exactly the RCCL symbols libweldacs.so imports: a new ncclXxx call in csrc/host_comm.inc cannot slip past the multi-rank tests"""
    import test_gpu_mock_ranks as M
    mock = M.build_mock()

    def syms(path, kind):
        r = subprocess.run(["nm", "-D", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return {l.split()[-1] for l in r.stdout.splitlines() if (" %s nccl" % kind) in l}
    wanted, have = syms(_lib.LIB_PATH, "U"), syms(mock, "T")
    assert len(wanted) >= 12 and wanted <= have, wanted - have


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
    # the owner word: which slot of which rank holds the path behind the global best (ties: the lower slot)
    cost, rk, sl = comm.read_best_owner(0, K)
    assert np.array_equal(bits(cost), bits(glob)) and np.all(rk == 0)
    assert np.array_equal(sl, np.where(bits(t1) < bits(t0), 1, 0))
    comm.close(); s.close(); dg.close(); ctx.close()


@pytest.mark.gpu
def test_only_exchanged_generations_can_be_read():
    """a generation inside the allocated history that never went through wa_acs_allreduce_best is refused (it used to come back as
    uninitialised device memory with WA_OK)"""
    ctx = api.Context(0)
    og = O.synth_grid(24, seed=5, occ_prob=0.1)
    free = np.nonzero(og.free)[0]
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    s = api.AcsSolver(ctx, dg, n_slots=1, max_colony=16)
    comm = api.Comm(ctx, 0, 1, api.Comm.unique_id())
    p = api.default_params(max_iteration=40, predict=60.0, fixed_colony=16, rng_mode=api.RNG_DEV, seed=2)
    s.init_pheromone(1.0)
    s.begin(p, int(free[0]), int(free[-1]))
    s.run(40)
    comm.allreduce_best(s, 10, 5)                    # generations 10..14 only
    assert comm.read_best(10, 5).shape == (5,)
    for g0, cnt in ((0, 5), (9, 2), (14, 2), (15, 1), (2 ** 31 - 2, 2)):
        with pytest.raises(api.WeldacsError):
            comm.read_best(g0, cnt)
    comm.allreduce_best(s, 0, 10)
    assert np.array_equal(bits(comm.read_best(0, 15)), bits(s.trace()["bestL"][:15]))
    comm.close(); s.close(); dg.close(); ctx.close()


@pytest.mark.gpu
def test_costs_and_paths_of_a_sharded_pair_run_reach_the_stitching_rank():
    """SURVEY 8(e)'s end-of-run exchanges behind the C ABI with the ranks a 1-GPU box has: wa_comm_allgather_costs (size-prefixed
    ncclAllGather) and wa_comm_gather_paths (size-prefixed gather to the root) are really issued over the world of one rank."""
    ctx = api.Context(0)
    comm = api.Comm(ctx, 0, 1, api.Comm.unique_id())
    rs = np.random.RandomState(4)
    idx = rs.permutation(40)[:17].astype(np.int32)
    cost = rs.uniform(1, 500, 17).astype(np.float32)
    cost[3] = np.inf                                  # "no path" is a valid cost
    allc = comm.allgather_costs(idx, cost, 40)
    want = np.full(40, np.nan, np.float32)
    want[idx] = cost
    assert np.array_equal(bits(allc), bits(want))
    assert comm.allgather_costs([], [], 5).shape == (5,) and np.isnan(comm.allgather_costs([], [], 5)).all()
    with pytest.raises(api.WeldacsError):
        comm.allgather_costs([40], [1.0], 40)
    paths = {int(k): rs.randint(0, 10 ** 6, int(rs.randint(0, 300))).astype(np.int32) for k in idx}
    got = comm.gather_paths(paths, root=0)
    assert set(got) == set(paths) and all(np.array_equal(got[k], paths[k]) for k in paths)
    assert comm.gather_paths({}, root=0) == {}
    comm.close(); ctx.close()


@pytest.mark.gpu
def test_grid_broadcast_with_one_rank_issues_the_collective_and_keeps_the_roots_grid():
    """wa_comm_broadcast_grid with the world a 1-GPU box can form: the header all-gather and the four ncclBroadcast calls are really issued
    (RCCL, in place), the root gets no replica (it keeps its own grid), bad arguments fail instead of hanging.  World > 1: the mock-ranks test."""
    ctx = api.Context(0)
    comm = api.Comm(ctx, 0, 1, api.Comm.unique_id())
    og = O.synth_grid(24, seed=5, occ_prob=0.2)
    dg = api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)
    before = dg.occupancy().copy()
    bg = comm.broadcast_grid(dg, root=0)
    assert bg is dg and np.array_equal(dg.occupancy(), before) and dg.n_free == int(og.free.sum())
    with pytest.raises(api.WeldacsError):
        comm.broadcast_grid(None, root=0)
    with pytest.raises(api.WeldacsError):
        comm.broadcast_grid(dg, root=3)
    assert comm.allreduce([2.5], "sum").tolist() == [2.5]      # the communicator still works
    comm.close(); dg.close(); ctx.close()


def test_best_key_packs_cost_and_owner_in_one_orderable_word():
    """host-side helpers of the global-best exchange (no device work): the key orders by cost first, then by rank, then by slot"""
    k = api.pack_best_key
    assert k(1.5, 0, 0) < k(2.5, 0, 0) < k(np.inf, 0, 0) and k(0.0, 3, 7) < k(1e-38, 0, 0)
    assert k(7.0, 1, 5) < k(7.0, 2, 0) and k(7.0, 2, 0) < k(7.0, 2, 1)
    assert api.unpack_best_key(k(378.0, 6, 123)) == (np.float32(378.0), 6, 123)
    c, r, s_ = api.unpack_best_key(k(np.inf, 7, 65535))
    assert np.isinf(c) and (r, s_) == (7, 65535)
    for bad in ((-1.0, 0, 0), (np.nan, 0, 0), (1.0, -1, 0), (1.0, 0, 65536), (1.0, 32768, 0)):
        with pytest.raises(api.WeldacsError):
            k(*bad)


@pytest.mark.gpu
def test_cpp_multistart_host_equals_the_oracle_trace():
    """examples/multistart_rccl.cpp (C4 from a C++ host: thread + wa_ctx + wa_comm per device) on the devices this box has:
    every rank's local history equals the oracle's DEV-mode run of that rank's problem, the global history is their MIN."""
    check_multistart("all")


def check_multistart(devices, env=None, want_ranks=None):
    """(also run by tests/test_gpu_mock_ranks.py with three ranks on one GPU, RCCL replaced by the mock)"""
    compile_multistart()
    n, ants, K = 48, 64, 60
    out = TMPW + "weldacs_ms_%d.txt" % os.getpid()
    r = subprocess.run([EXE, str(n), str(ants), str(K), devices, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout + r.stderr
    loc, glob, owner, owner_path = {}, {}, {}, None
    for line in open(out):
        t = line.split()
        if t[0] == "local":
            loc.setdefault(int(t[1]), {})[int(t[2])] = int(t[3], 16)
        elif t[0] == "global":
            glob[int(t[1])] = int(t[2], 16)
            owner[int(t[1])] = (int(t[3]), int(t[4]))
        elif t[0] == "owner_path":
            owner_path = (int(t[1]), [int(v) for v in t[3:3 + int(t[2])]])
    W = len(loc)
    assert W >= 1 and len(glob) == K and (want_ranks is None or W == want_ranks)
    hist, best_paths = [], []
    for rk in range(W):
        og = O.synth_grid(n, seed=2024 + rk, occ_prob=0.10)
        free, *_ = synth.synth_grid(n, seed=2024 + rk, occ_prob=0.10)
        assert np.array_equal(free, og.free)
        sid, eid = og.resolve(np.zeros(3, np.float32)), og.resolve(np.full(3, n - 1, np.float32))
        a = O.Acs(og)
        tr = a.solve(sid, eid, K, float(np.float32(ants / 0.35)), fixed_colony=ants, mode=O.DEV, seed=12345 + rk, stream=rk)
        assert [loc[rk][g] for g in range(K)] == bits(tr["bestL"]).tolist(), rk
        hist.append(tr["bestL"])
        best_paths.append(a.best_path()[0])
    H = np.stack(hist)
    assert [glob[g] for g in range(K)] == bits(np.min(H, 0)).tolist()
    # the owner the packed key carries: the lowest rank among those that hold the minimum, slot 0; and its path reached rank 0
    assert [owner[g] for g in range(K)] == [(int(np.argmin(H[:, g])), 0) for g in range(K)]
    assert owner_path is not None and owner_path[0] == owner[K - 1][0] and owner_path[1] == best_paths[owner_path[0]].tolist()
