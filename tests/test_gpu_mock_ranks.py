"""The multi-rank paths of the library's communicator (wa_comm_*, csrc/host_comm.inc: SURVEY 8(e)'s three exchanges) with world = 2 and 3
on a box that has ONE GPU.  RCCL refuses two ranks on one device, so the fifteen RCCL entry points the library calls are replaced by
tests/mock_rccl (LD_PRELOAD; file exchange between processes that share the GPU): what runs is every line of the library around the
collectives -- packed keys and owners, size prefixes, padding, offsets at a root that is not rank 0, ragged and empty contributions --
which a world of one rank (tests/test_comm.py, the most a 1-GPU box offers RCCL itself) never exercises.  Not a test of RCCL."""
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest
from tmpw import TMPW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
MOCK = TMPW + "weldacs_libmock_rccl_%d.so" % os.getuid()
MOCK_DIR = TMPW + "weldacs_mock_rccl_%d" % os.getuid()


def build_mock():
    r = subprocess.run(["g++", "-std=c++14", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"),
                        "-L/opt/rocm/lib", "-lamdhip64", "-o", MOCK], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return MOCK


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, argv, timeout=240, extra_env=None):
    """argv as `world` processes that share GPU 0, the mock in front of librccl; returns their outputs"""
    build_mock()
    shutil.rmtree(MOCK_DIR, ignore_errors=True)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LD_PRELOAD=MOCK, MOCK_RCCL_DIR=MOCK_DIR, MOCK_RCCL_TIMEOUT_S="60", WA_BENCH_BACKEND="gloo")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    shutil.rmtree(MOCK_DIR, ignore_errors=True)
    for rc, o, e in outs:
        assert rc == 0, o[-1500:] + e[-3000:]
    return outs


def key(cost, rank, slot):
    return (int(np.float32(cost).view(np.uint32)) << 32) | (rank << 16) | slot


@pytest.mark.parametrize("world", [2, 3])
def test_every_exchange_of_the_communicator_between_ranks(world):
    out = TMPW + "weldacs_mock_ranks_%d" % world
    run_ranks(world, [os.path.join(ROOT, "tests", "mock_rccl", "ranks.py"), out])
    R = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
    K = R[0]["cost"].size
    # (1) the global best of every generation is the MIN over ranks and slots of the local bests, and its owner is who holds it
    #     (ties: lowest rank, then lowest slot -- the order of the packed key)
    for g in range(K):
        best = min((key(R[r]["mine"][q][g], r, q), r, q) for r in range(world) for q in range(2))
        for r in range(world):
            assert key(R[r]["cost"][g], int(R[r]["owner_rank"][g]), int(R[r]["owner_slot"][g])) == best[0], (g, r)
    assert len({int(R[0]["owner_rank"][g]) for g in range(K)}) >= 1
    # the ranks really ran different searches (otherwise the owner would always be rank 0 for a trivial reason)
    assert any(not np.array_equal(R[0]["mine"], R[r]["mine"]) for r in range(1, world))
    # (2) host-side reductions
    for r in range(world):
        assert np.array_equal(R[r]["red_min"], [1.5, -(world - 1.0), 2.0])
        assert np.array_equal(R[r]["red_max"], [world + 0.5, 0.0, 2.0])
        assert np.array_equal(R[r]["red_sum"], [sum(k + 1.5 for k in range(world)), -sum(range(world)), 2.0 * world])
    # (3) every rank holds every pair cost; the two entries nobody owns keep the fill value
    want = np.full(sum(range(world)) + 2, -1.0, np.float32)
    for r in range(world):
        for i in range(r):
            want[sum(range(r)) + i] = 100.0 * r + i + 0.25
    for r in range(world):
        assert np.array_equal(R[r]["vec"], want), r
    # (4) the root (the LAST rank) holds every path of every rank, the others nothing
    root = world - 1
    sent = {}
    for r in range(world):
        off = np.concatenate([[0], np.cumsum(R[r]["sent_lens"])])
        for i, k in enumerate(R[r]["sent_keys"]):
            sent[int(k)] = R[r]["sent_ids"][off[i]:off[i + 1]]
    assert len(sent) == sum(r + 2 for r in range(world - 1)) and any(len(v) == 0 for v in sent.values())
    for r in range(world):
        if r != root:
            assert R[r]["got_keys"].size == 0
            continue
        off = np.concatenate([[0], np.cumsum(R[r]["got_lens"])])
        got = {int(k): R[r]["got_ids"][off[i]:off[i + 1]] for i, k in enumerate(R[r]["got_keys"])}
        assert sorted(got) == sorted(sent)
        for k in sent:
            assert np.array_equal(got[k], sent[k]), k


    # (5) every rank holds the ROOT's grid (rank 1 voxelised the mesh on the device), and that grid is the oracle's (model_grid_map.hpp:151-298)
    import oracle_lib as O
    og = O.grid_from_mesh(O.stl_parse(open(os.path.join(ROOT, "tests", "golden", "cubic.stl"), "rb").read()), 0.0219, 8)
    for r in range(world):
        assert np.array_equal(R[r]["b_occ"], og.free), r
        for a, b in ((R[r]["b_cx"], og.cx), (R[r]["b_cy"], og.cy), (R[r]["b_cz"], og.cz)):
            assert np.array_equal(a.view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)), r
        assert R[r]["b_meta"].tolist() == [og.nx, og.ny, og.nz, 8, int(og.free.sum()), int(np.float32(0.0219).view(np.uint32))], r
        assert R[r]["b_ids"].tolist() == [4130, 17521], r                     # KA2's start and end voxel
    # (6) bad arguments on ONE rank: every rank returned an error from that call (nobody hung), and the communicator went on working
    for r in range(world):
        assert R[r]["errs"].tolist() == [1, 1, 1], (r, R[r]["errs"])
        assert R[r]["after"].tolist() == [float(sum(range(world)))]


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_that_fails_inside_an_exchange_and_a_rank_that_leaves(world):
    """VERDICT r05 task 6: one rank's staging allocation fails INSIDE wa_comm_gather_paths / wa_comm_allgather_costs (behind the first header
    round; forced through the knobs build): every rank gets WA_ERR_ALLOC from that very call, nobody waits for a partner that has left it,
    and the communicator goes on working; a rank that exits without a word makes its peers' next exchange end with an error in bounded
    time; an aborted communicator refuses further calls with WA_ERR_STATE."""
    from welding_robot_amd import build as wb
    knobs = wb.build_knobs()
    out = TMPW + "weldacs_mock_fail_%d" % world
    run_ranks(world, [os.path.join(ROOT, "tests", "mock_rccl", "ranks.py"), out], extra_env=dict(RANKS_MODE="fail_inside", WELDACS_LIB=knobs, MOCK_RCCL_TIMEOUT_S="4"))
    R = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
    from welding_robot_amd._lib import STATUS
    code = {v: k for k, v in STATUS.items()}
    for r in range(world):
        assert R[r]["errs"].tolist() == [code["WA_ERR_ALLOC"], code["WA_ERR_ALLOC"]], (r, R[r]["errs"])
        assert R[r]["after"].tolist() == [float(sum(range(world)))]
        assert R[r]["vec"].tolist() == [1.0 + q for q in range(world)]
        assert int(R[r]["n_got"][0]) == (world if r == 0 else 0)
        if r != world - 1:
            assert int(R[r]["dead_code"][0]) == code["WA_ERR_DEVICE"] and 0 < float(R[r]["t_dead"][0]) < 30, (r, R[r]["t_dead"], R[r]["dead_code"])
            assert int(R[r]["ab"][0]) == code["WA_ERR_STATE"]


@pytest.mark.parametrize("world", [2, 3])
def test_pair_planning_sharded_over_ranks_equals_the_single_process_run(world):
    """examples/plan_batch.py (BASELINE config C5 in miniature) as `world` processes: pairs dealt longest-first over the ranks, costs all-gathered,
    paths gathered to rank 0, which orders the seams and stitches -- the same tour, the same stitched path and the same trajectory as one process"""
    args = [os.path.join(ROOT, "examples", "plan_batch.py"), "--grid", "48", "--points", "9", "--generations", "40", "--slots", "6"]
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    outs = run_ranks(world, args)
    many = json.loads([l for l in outs[0][1].splitlines() if l.startswith("{")][0])
    assert all(not [l for l in o.splitlines() if l.startswith("{")] for _, o, _ in outs[1:])     # only rank 0 reports
    assert many["world"] == world and many["pairs_this_rank"] < one["pairs_this_rank"] == 36
    for k in ("all_reached", "tour_cost", "tour_iterations", "order", "stitched_nodes", "coarse_points", "trajectory_samples", "trajectory_length"):
        assert many[k] == one[k], k


def test_cpp_multistart_host_with_three_ranks_on_one_gpu():
    """examples/multistart_rccl.cpp with the device list "0,0,0": three host threads, each with its own wa_ctx + wa_comm + search (BASELINE config C4's
    topology inside one process), the mock between them: every rank's history equals the oracle's run of that rank's problem, the global history is
    their MIN with the owner the packed key names, and the OWNER'S best path -- fetched from the rank that holds it -- is that rank's oracle path."""
    import test_comm
    build_mock()
    shutil.rmtree(MOCK_DIR, ignore_errors=True)
    test_comm.check_multistart("0,0,0", env={"LD_PRELOAD": MOCK, "MOCK_RCCL_DIR": MOCK_DIR, "MOCK_RCCL_TIMEOUT_S": "60"}, want_ranks=3)
    shutil.rmtree(MOCK_DIR, ignore_errors=True)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_with_ranks_on_one_gpu(world):
    """bench.py --gpus N as the round driver launches it (RANK / WORLD_SIZE / MASTER_* in the environment), all ranks on the one GPU: the library's
    exchange through the mock, torch's own barrier and small reductions over gloo (WA_BENCH_BACKEND=gloo; on a multi-GPU node both are RCCL).
    bench.py itself asserts that the reduced global-best history is the MIN of the ranks' gathered histories; here: one JSON line, from rank 0,
    that counts every rank's generations and names the owner of the final global best -- and the C5 STRONG-scaling leg (`c5_sharded`: the grid
    broadcast from rank 0, the pairs dealt over the ranks, costs all-gathered, paths gathered, the seam order on rank 0), shrunk to fit N
    processes on one GPU, which bench.py holds against the one-rank run of the same job."""
    outs = run_ranks(world, [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "30", "--warmup", "3"], timeout=400,
                     extra_env={"WA_BENCH_C5": "48,9,40", "WA_BENCH_C5_SLOTS": "6"})
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not any(l.startswith("{") for _, o, _ in outs[1:] for l in o.splitlines())
    d = json.loads(lines[0])
    c5 = d["c5_sharded"]
    assert c5["scaling"] == "strong" and c5["ranks"] == world and c5["pairs"] == 36 and c5["paths_on_rank0"] == 36 and c5["equals_one_rank_run"] is True
    assert c5["all_reached"] and 0 < c5["pairs_per_rank_mean"] < 36 and c5["t_search_s"]["imbalance_max_over_mean"] >= 1.0 and c5["pair_generations_per_s"] > 0
    if world != 2:
        assert d["n_gpus"] == world and d["scaling"] == "weak"
        return
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["scaling"] == "weak" and "wa_acs_allreduce_best" in d["config"]["global_best_allreduce"]
    assert abs(d["value"] - 2 * 30 / (d["ms_per_step"] * 30 * 1e-3)) < 1e-6 * d["value"]      # whole-job rate: both ranks' generations over the slowest rank's time
    assert d["global_best_owner"][0] in (0, 1) and d["global_best_owner"][1] == 0
    assert d["best_cost_all_ranks"] <= d["best_cost"]
    assert "cpu_baseline" not in d and "multi_start" not in d          # N > 1: the timed job only


def test_bench_survives_a_rank_that_cannot_plan_its_share():
    """one rank of the sharded C5 leg fails before the exchange (out of device memory, say): the ranks agree to give the leg up -- nobody waits
    in a collective for the rank that left --, rank 0 still prints the line with the headline, and the failure is reported under c5_sharded"""
    outs = run_ranks(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2"], timeout=300,
                     extra_env={"WA_BENCH_C5": "48,9,40", "WA_BENCH_C5_SLOTS": "6", "WA_BENCH_FAIL_RANK": "1"})
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "another rank" in d["c5_sharded"]["error"]
    assert "WA_BENCH_FAIL_RANK" in outs[1][2]       # rank 1 said why on its stderr
