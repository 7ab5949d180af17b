"""BASELINE configs at FULL size on the GPU: C3 (128^3, 256 ants) deep into convergence against the
oracle and the reference; C5 (256^3, 64 weld points, 2 016 pair searches x 150 generations + the 64-seam order);
C4's code path (RCCL MIN all-reduce of the per-generation best) with the one rank a 1-GPU box has."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import waf
from welding_robot_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def dgrid(ctx, og):
    return api.Grid.from_occupancy(ctx, og.free, og.cx, og.cy, og.cz, og.precision, og.wall)


# ------------------------------------------------------------------ C3
def test_c3_dev_150_generations_trace_path_and_whole_field_equal_the_oracle(ctx):
    """Exploration (generations 0-30), the mixed regime where most ants replay a prefix of the best path and finish
    in the general loop (30-80), and the converged regime (all 256 ants on the replay track, 50 ranks depositing on
    one path): every generation's trace, the best path and all 12.6 M pheromone values, bit for bit."""
    og = O.synth_grid(128, seed=2024, occ_prob=0.10)
    s = api.AcsSolver(ctx, dgrid(ctx, og), n_slots=1, max_colony=256)
    p = api.default_params(max_iteration=150, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.solve(p, 16513, 2097151)
    a = O.Acs(og)
    tr = a.solve(16513, 2097151, 150, 731.43, fixed_colony=256, mode=O.DEV, seed=12345, stream=0)
    t = s.trace()
    for k in ("bestL", "iterbestL"):
        assert np.array_equal(bits(t[k]), bits(tr[k])), k
    for k in ("colony", "finite", "steps"):
        assert np.array_equal(t[k], tr[k]), k
    cost, path, ch = s.result()
    ids, och = a.best_path()
    assert bits(cost) == bits(a.best_L) and float(cost) == 378.0            # the Manhattan optimum 3 x 126
    assert np.array_equal(path, ids) and np.array_equal(ch.astype(np.int32), och)
    L, lens = s.ants()
    olens, oL = a.last_ants()
    assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))
    assert t["steps"][-1] == 256 * 378                                        # converged: every ant walks the best path
    assert np.array_equal(bits(s.pheromone()), bits(a.pheromone()))
    s.close()


def test_c3_dev_500_generations_at_its_stated_length_equal_the_oracle(ctx):
    """BASELINE config 3 at its stated 500 iterations (ACSRank_3D.hpp:237-299): the fused sweep + rank + mark launch and
    the 50-rank deposits carry the fp32 field through the denormal range (off-path edges: 1 * 0.8^t goes subnormal at
    t ~ 392, :268-272) and onto the 2-ulp fixed point (t ~ 460).  Every generation's trace, the best path,
    every ant's (L, length) and all 12.6 M pheromone values against the oracle, bit for bit."""
    og = O.synth_grid(128, seed=2024, occ_prob=0.10)
    s = api.AcsSolver(ctx, dgrid(ctx, og), n_slots=1, max_colony=256)
    p = api.default_params(max_iteration=500, predict=731.43, fixed_colony=256, rng_mode=api.RNG_DEV, seed=12345)
    s.solve(p, 16513, 2097151)
    a = O.Acs(og)
    tr = a.solve(16513, 2097151, 500, 731.43, fixed_colony=256, mode=O.DEV, seed=12345, stream=0)
    t = s.trace()
    assert len(t["bestL"]) == 500
    for k in ("bestL", "iterbestL"):
        assert np.array_equal(bits(t[k]), bits(tr[k])), k
    for k in ("colony", "finite", "steps"):
        assert np.array_equal(t[k], tr[k]), k
    cost, path, ch = s.result()
    ids, och = a.best_path()
    assert bits(cost) == bits(a.best_L) and float(cost) == 378.0
    assert np.array_equal(path, ids) and np.array_equal(ch.astype(np.int32), och)
    L, lens = s.ants()
    olens, oL = a.last_ants()
    assert np.array_equal(lens, olens) and np.array_equal(bits(L), bits(oL))
    f, of = s.pheromone(), a.pheromone()
    assert np.array_equal(bits(f), bits(of))
    # the regime this test is about really was reached: off-path in-bounds edges sit on the denormal fixed point
    tiny = f[(f > 0) & (f < np.float32(1.1754944e-38))]
    assert tiny.size > 6 * 128 ** 3 // 2 and bits(tiny.min()) == 2
    s.close()


@pytest.mark.parametrize("gens", [40, 500])
def test_c3_ref_generations_equal_the_reference(ctx, gens):
    """REF mode (libc stream, std::sort ties) against the reference's own 40 generations on the C3 grid and against its
    full 500 (BASELINE config 3 as stated; denormal range and 2-ulp fixed point of the field included): trace, path,
    field hash and the position of the libc stream."""
    g = waf.load(os.path.join(G, "acs_synth128_fixed256_%d.waf" % gens))
    og = O.synth_grid(128, seed=2024, occ_prob=0.10)
    s = api.AcsSolver(ctx, dgrid(ctx, og), n_slots=1, max_colony=256)
    s.srand(12345)
    p = api.default_params(max_iteration=gens, predict=float(np.float32("731.43")), fixed_colony=256, rng_mode=api.RNG_REF)
    s.solve(p, 16513, 2097151)
    cost, path, ch = s.result()
    assert bits(cost) == bits(g["best_L"]) and np.array_equal(path, g["best_path"]) and np.array_equal(ch.astype(np.int32), g["best_choice"])
    t = s.trace()
    assert np.array_equal(bits(t["bestL"]), bits(g["tr_bestL"])) and np.array_equal(bits(t["iterbestL"]), bits(g["tr_iterbestL"]))
    assert np.array_equal(t["finite"], g["tr_finite"]) and np.array_equal(t["steps"], g["tr_steps"])
    assert O.pher_hash(s.pheromone()) == waf.scalar(g, "pher_hash") & ((1 << 64) - 1)
    st = s.rand_state()
    rng = O.GlibcRand()
    for i in range(34):
        rng.r[i] = int(st[i])
    rng.f, rng.b = int(st[34]), int(st[35])
    assert O.rand(rng) == waf.scalar(g, "next_rand")
    s.close()


# ------------------------------------------------------------------ C5
def test_c5_full_size_pair_planning_and_seam_order(ctx):
    """256^3 synthetic grid (seed 2024), 64 weld points, ALL 2 016 pair searches x 150 generations with lazy evaporation
    (48 concurrent slots), then the 64-seam order.  Checked: a 32-pair slice with the dense 805 MB sweep gives the same
    costs and paths; two pairs equal the oracle generation by generation; the ACS-TSP tour equals the oracle's on the
    same matrix; the tour is a Hamiltonian cycle over the 64 points."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("plan_batch", os.path.join(ROOT, "examples", "plan_batch.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    n, P, gens, seed = 256, 64, 150, 7
    predict = float(0.35 ** -1 * 24)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    dg = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    assert dg.n == 16777216
    pts = synth.synth_weld_points(free, n, P, seed=seed)
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    assert len(pairs) == 2016
    cost, paths, mine = pb.plan(ctx, dg, pts, gens, predict, seed, slots=48, lazy=True)
    assert mine == 2016 and np.isfinite(cost).all()
    # dense sweep on a 32-pair slice (pairs 0..31 of the global numbering: stream keys are global pair indices)
    sl = pairs[:32]
    solver = api.AcsSolver(ctx, dg, n_slots=32, max_colony=max(1, int(0.35 * predict)))
    p = api.default_params(max_iteration=gens, predict=predict, rng_mode=api.RNG_DEV, seed=seed)
    solver.solve(p, [pts[i] for i, _ in sl], [pts[j] for _, j in sl], streams=list(range(32)))
    dc, dp = solver.results(32)
    for q, (i, j) in enumerate(sl):
        assert bits(np.float32(cost[i, j])) == bits(dc[q]), (i, j)
        assert np.array_equal(dp[q], paths[(i, j)]), (i, j)
    solver.close()
    # two pairs against the oracle, every generation of a shorter run (same stream keys => same draws)
    og = O.Grid(cx, cy, cz, free, 1.0, 0)
    for lazy in (False, True):
        s2 = api.AcsSolver(ctx, dg, n_slots=2, max_colony=max(1, int(0.35 * predict)), lazy=lazy)
        ks = [5, 1000]
        p8 = api.default_params(max_iteration=8, predict=predict, rng_mode=api.RNG_DEV, seed=seed)
        s2.solve(p8, [pts[pairs[k][0]] for k in ks], [pts[pairs[k][1]] for k in ks], streams=ks)
        for q, k in enumerate(ks):
            a = O.Acs(og)
            tr = a.solve(int(pts[pairs[k][0]]), int(pts[pairs[k][1]]), 8, predict, mode=O.DEV, seed=seed, stream=k)
            t = s2.trace(q)
            assert np.array_equal(bits(t["bestL"]), bits(tr["bestL"])) and np.array_equal(t["steps"], tr["steps"]), (lazy, k)
            assert np.array_equal(t["colony"], tr["colony"]) and np.array_equal(t["finite"], tr["finite"])
            c, ids, _ = s2.result(q)
            assert bits(c) == bits(a.best_L) and (not np.isfinite(c) or np.array_equal(ids, a.best_path()[0]))
            if k == 5:
                assert np.array_equal(bits(s2.pheromone(q)), bits(a.pheromone())), (lazy, k)   # 100 M values
            del a
        s2.close()
    # the weld-seam order from the in-memory matrix (no graph.in round trip, Q6)
    tour = api.gtsp_solve(ctx, cost, mode=api.RNG_DEV, seed=seed)
    o = O.gtsp_solve(cost, mode=O.DEV, seed=seed)
    assert tour["L"][0] == o["L"] and np.array_equal(tour["edges"][0], o["edges"]) and int(tour["iters"][0]) == o["iters"]
    order = [int(e[0]) for e in tour["edges"][0]]
    assert sorted(order) == list(range(P))
    # the context keeps the blocks of the solvers destroyed above for its next solver (wa_ctx_cached_bytes); the tests below start OTHER
    # processes on this GPU, which would wait for that memory (bench.py: wait_for_device_memory)
    # (round 5: the arena builds every solver from the same chunks, so what is kept is the largest footprint -- the 48-slot solver's ~41 GB
    #  -- not the sum of every shape's blocks as with round 4's exact-fit cache)
    assert ctx.cached_bytes() > (30 << 30)
    ctx.trim()
    assert ctx.cached_bytes() == 0


# ------------------------------------------------------------------ C4 with the ranks a 1-GPU box has
def _free_port():
    """a port nobody listens on right now (a fixed one can still be in TIME_WAIT from the test before)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(args, env=None, launcher=()):
    cmd = list(launcher) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})), cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("path", ["libweldacs", "torch"])
def test_c4_rccl_path_with_one_rank(path):
    """`torchrun --nproc-per-node 1` + WA_FORCE_DIST=1: a process group of ONE rank, which is all a 1-GPU box can form.  The
    collective is still ISSUED: wa_acs_allreduce_best = k_min_over_slots + ncclAllReduce(ncclMin) on the communicator's stream
    (default), or torch.distributed.all_reduce(MIN) of the exported trace (WA_BENCH_TORCH_ALLREDUCE=1); bench.py asserts that
    the reduced history equals the rank's own.  What this does NOT show is a reduction over more than one rank's data: N > 1
    is covered by the 2-rank gloo test on the CPU and is unmeasured on hardware."""
    port = _free_port()
    env = {"WA_FORCE_DIST": "1"}
    if path == "torch":
        env["WA_BENCH_TORCH_ALLREDUCE"] = "1"
    d = _bench(["--gpus", "1", "--steps", "100", "--warmup", "5", "--no-cpu", "--no-extras"], env=env,
               launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)])
    assert d["n_gpus"] == 1 and d["steps"] == 100 and d["config"]["global_best_allreduce"].startswith("RCCL MIN")
    assert ("wa_acs_allreduce_best" in d["config"]["global_best_allreduce"]) == (path == "libweldacs")
    assert d["best_cost"] == d["best_cost_all_ranks"] and np.isfinite(d["best_cost"]) and d["value"] > 0


def test_c5_sharded_leg_with_one_rank_over_real_rccl():
    """bench.py's C5 strong-scaling leg (`c5_sharded`) with the one rank RCCL itself admits on a 1-GPU box: the grid broadcast, the cost all-gather
    and the path gather are ISSUED through librccl (no stand-in), and the result equals the plain one-rank plan.  (Several ranks: the mock-ranks test.)"""
    port = _free_port()
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "2", "--no-cpu", "--no-extras"], env={"WA_FORCE_DIST": "1", "WA_BENCH_C5": "96,16,80"},
               launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)])
    c5 = d["c5_sharded"]
    assert c5["ranks"] == 1 and c5["pairs"] == 120 and c5["paths_on_rank0"] == 120 and c5["equals_one_rank_run"] is True and c5["all_reached"]


def test_c4_workload_of_rank_7_on_this_gpu():
    """Rank 7's C4 problem (grid seed 2031, colony seed 12352) run here: the DEV-mode port draws the same numbers, so the
    per-generation best-cost trace must be bit-equal to the CPU's."""
    d = _bench(["--steps", "60", "--warmup", "2", "--workload-index", "7", "--cpu-gens", "12", "--no-extras"])
    assert "seed 2024+rank" in d["config"]["workload"] and d["cost_check"]["bit_equal_trace"] is True
    assert d["cost_check"]["generations"] == 60 and d["cost_check"]["best_path_equal"] is True and d["cpu_baseline"]["value"] > 0
