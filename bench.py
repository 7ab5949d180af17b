#!/usr/bin/env python3
"""bench.py -- ACS generations/sec on the BASELINE workload (config C3 per GPU; C4 across GPUs).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE ACS generation (walk + rank + evaporate + ranked deposit, ACSRank_3D.hpp:237-299)
of the 128^3 / 256-ant search.  Every rank owns an independent 128^3 random-obstacle grid (weak
scaling, no data-path collective); for N > 1 the per-generation global-best cost is MIN
all-reduced over RCCL in chunks of generations, overlapped with the next chunk.  W warm-up
generations run on a throw-away search; then EXACTLY K generations of a fresh search are timed
between barrier + synchronize pairs, inputs already resident in HBM (the per-problem setup --
pheromone init, heuristic field -- is outside the timed region and reported as setup_ms).

Rank 0 prints one JSON line: metric/value (whole-job generations/s), `roofline` for the
evaporation sweep and -- at N = 1 -- `cpu_baseline`: the reference's own loop (oracle/_ref/ref_harness,
kind "reference") or the C oracle (kind "port") timed on this host on a bounded sample.

roofline: the launch of the timed loop that carries the evaporation sweep (ACSRank_3D.hpp:268-272: 48 B of
algorithmic traffic per voxel) is `k_evap_rank_mark` -- 4096 sweep blocks + the latency-bound rank / mark blocks of the
same generation.  Its launches are stamped per dispatch with HIP events on the library's own stream (hipExtLaunchKernelGGL
start/stop) over the timed region; `achieved` = 48 N^3 / that average, `frac` against the 8 TB/s HBM3E peak; `traffic` = HBM bytes per
launch from two rocprofv3 PMC child passes of this very run (FETCH_SIZE / WRITE_SIZE; the committed passes when that is not possible).  At 128^3
both 48 MiB buffers sit in the 256 MiB Infinity Cache, so beside it `sweep_alone` reports the sweep kernel itself
(`k_evaporate`, 64 stamped launches after the timed region) at 128^3 and at 256^3 (805 MB per launch: past the cache).
Extras after the timed region, outside `value`: `full_run` (the same search over BASELINE config 3's stated 500
generations), `multi_start` (eight independent searches advancing together on this GPU: config 4's
workload, with an HBM-resident sweep), `c5_full` (BASELINE config 5 at full size on this GPU), `walk_step` (time per general step of the walk
against the issue floor of its assembled loop), `c5_pair_planning` (dense vs lazy evaporation, must agree).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GRID_N, ANTS, CHUNK = 128, 256, 50
PREDICT = 731.43  # = 256 ants * precision / 0.35 (ACSRank_3D.hpp:247); only scales Q until a first path exists


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=GRID_N)
    ap.add_argument("--ants", type=int, default=ANTS)
    ap.add_argument("--cpu-gens", type=int, default=100, help="generations of the CPU baseline's second sample (window_100) when the timed region is shorter")
    ap.add_argument("--cpu-gens-max", type=int, default=500, help="the CPU baseline runs the timed region's generations, at most this many")
    ap.add_argument("--cost-check-gens", type=int, default=None, help="generations the CPU port replays for cost_check (default: all K)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--profile-every", type=int, default=10, help="stamp every n-th generation's launches with HIP events")
    ap.add_argument("--no-extras", action="store_true",
                    help="warm-up + timed region only: no continuation of the search behind the timed region (roofline samples), no walk_step, no secondary measurements "
                         "-- what a rocprofv3 kernel trace of the command should contain (tools/timed_window_stats.py)")
    ap.add_argument("--no-roofline-256", action="store_true", help="skip the 256^3 (past the Infinity Cache) sweep measurement")
    ap.add_argument("--workload-index", type=int, default=None,
                    help="run rank R's C4 workload (grid seed 2024+R, colony seed 12345+R) on this GPU; default = own rank")
    return ap.parse_args()


def _reference_run(O, og, n, wl, ants, gens, tmp):
    """the reference's own loop (oracle/_ref/ref_harness) on `gens` generations of the search: its JSON line or None"""
    p = subprocess.run([O.REF_BIN, "acs", "gridin=%s/grid.in" % tmp, "spt=0,0,0", "ept=%d,%d,%d" % (n - 1, n - 1, n - 1),
                        "seed=%d" % wl["rng_seed"], "iters=%d" % gens, "predict=%s" % PREDICT, "fixed=%d" % ants, "out=%s/o.waf" % tmp],
                       stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    line = [l for l in p.stderr.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if p.returncode == 0 and line else None


def cpu_baseline(args, free, n, gpu_trace, wl, gpu_path, K, gpu_window_ms):
    """Reported baseline (never the target).  Same grid, same parameters, THE SAME GENERATIONS as the timed region (0..K-1 of the same
    search; capped at --cpu-gens-max), 1 thread like the reference.  `window_100`: the first 100 generations as a second sample when the
    timed region is shorter (exploration and the start of convergence).
    cost_check: the DEV-mode port draws the same numbers as the GPU, so ALL K generations of the timed search are
    replayed on the CPU (66 s for 500) and the per-generation best cost, the iteration best, the step counts and the
    final best path must be equal bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O  # cpu_baseline leg only
    G = min(K, args.cpu_gens_max)
    og = O.Grid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), free, 1.0, 0)
    sid, eid = og.resolve(np.zeros(3, np.float32)), og.resolve(np.full(3, n - 1, np.float32))
    out = {}
    KC = K if args.cost_check_gens is None else min(K, args.cost_check_gens)
    a = O.Acs(og)
    t0 = time.time()
    tr = a.solve(sid, eid, KC, PREDICT, fixed_colony=args.ants, mode=O.DEV, seed=wl["rng_seed"], stream=wl["stream"])
    t_port = time.time() - t0
    port_rate = KC / t_port
    bit = lambda x: np.ascontiguousarray(x, np.float32).view(np.uint32)
    cost_equal = bool(np.array_equal(bit(tr["bestL"]), bit(gpu_trace["bestL"][:KC])) and
                      np.array_equal(bit(tr["iterbestL"]), bit(gpu_trace["iterbestL"][:KC])) and
                      np.array_equal(tr["steps"], gpu_trace["steps"][:KC]))
    path_equal = bool(np.array_equal(a.best_path()[0], gpu_path)) if KC == K else None
    out["cost_check"] = {"generations": KC, "of_timed_generations": K, "cpu_best_cost": float(tr["bestL"][-1]),
                         "gpu_best_cost": float(gpu_trace["bestL"][KC - 1]),
                         "bit_equal_trace": cost_equal, "best_path_equal": path_equal,
                         "checked": "best cost, iteration best and total steps of every generation; node ids of the final best path"}
    sample = "generations 0..%d of the same %d^3 / %d-ant search = %s; the GPU's timed region took %.2f ms for %d generations" % (
        G - 1, n, args.ants, "the timed region's generations" if G == K else "the first %d of the timed region's %d" % (G, K), gpu_window_ms, K)
    if O.have_ref():
        tmp = "/tmp/weld_bench_%d" % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        O.write_grid_in(og, tmp + "/grid.in")
        j = _reference_run(O, og, n, wl, args.ants, G, tmp)
        if j is not None:
            out["cpu_baseline"] = {"value": G / j["t_solve"], "unit": "generations/s", "cores": 1, "kind": "reference", "generations": [0, G - 1],
                                   "sample": sample + "; reference phase split walk %.0f%% evaporate %.0f%% deposit %.0f%%" % (
                                       100 * j["t_walk"] / j["t_solve"], 100 * j["t_evap"] / j["t_solve"], 100 * j["t_dep"] / j["t_solve"]),
                                   "init_s": j["t_init"], "port_value": port_rate, "port_generations": [0, KC - 1]}
            if G < 100 and args.cpu_gens > G:   # a second, longer sample of the same search
                j2 = _reference_run(O, og, n, wl, args.ants, args.cpu_gens, tmp)
                if j2 is not None:
                    out["cpu_baseline"]["window_%d" % args.cpu_gens] = {"value": args.cpu_gens / j2["t_solve"], "generations": [0, args.cpu_gens - 1], "unit": "generations/s"}
            return out
    out["cpu_baseline"] = {"value": port_rate, "unit": "generations/s", "cores": 1, "kind": "port", "generations": [0, KC - 1], "sample": sample}
    return out


def sweep_roofline(ctx, n, repeats=64):
    """k_evaporate alone on an n^3 field: 64 launches, each stamped by its own start/stop events."""
    import numpy as np
    from welding_robot_amd import api, synth
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    s = api.AcsSolver(ctx, grid, n_slots=1, max_colony=8, path_capacity=1024)
    s.init_pheromone(1.0)
    s.evaporate(0, 0.8, 4)           # warm-up launches (untimed)
    s.sync()
    s.profile(True, 1)
    s.evaporate(0, 0.999, repeats)
    pr = s.profile_read()["evaporate"]
    s.close()
    grid.close()
    ms = pr["ms"] / max(pr["launches"], 1)
    nbytes = 48.0 * n ** 3
    return {"grid": n, "launches": pr["launches"], "avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes,
            "achieved": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}


def _walk_loop_evidence():
    """instructions per step from the ISA dump tools/walk_isa.py writes at build time, cycles per instruction and shader clock of
    a lone wavefront from the issue-rate microbenchmark's committed output -- nothing of this is a constant in this file"""
    import glob
    import re
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "walk_loop_isa.txt")), reverse=True):
        head = open(f).read(600)
        m = re.match(r"# instructions_per_step: (\d+)", head)
        if m:
            out["instructions_per_step"], out["isa_file"] = int(m.group(1)), os.path.relpath(f, ROOT)
            m2 = re.search(r"# vector_memory_per_step: (\d+)\s+conditional_branches_per_step: (\d+)", head)
            if m2:
                out["vector_memory_per_step"], out["conditional_branches_per_step"] = int(m2.group(1)), int(m2.group(2))
            break
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "issue_rates.txt")), reverse=True):
        m = re.search(r"v_add_f32 dependent\s+([\d.]+) ticks.*?=\s*([\d.]+) GHz", open(f).read())
        if m:
            out["cycles_per_instruction"], out["shader_clock_ghz"], out["issue_rates_file"] = float(m.group(1)), float(m.group(2)), os.path.relpath(f, ROOT)
            break
    return out


def walk_step_extra(solver, p, ids, wl, generations=12):
    """What the exploratory generations are made of, measured after the timed region on a fresh identical search: a walk
    launch lasts as long as its slowest ant, so launch time / the longest walk of the generation = the time of one
    general step of a lone wavefront.  `issue_floor_ns` = instructions per step (from the ISA dump of this build's loop) x
    the issue cost of one instruction of a lone wavefront (microbenchmark): the bound this issue-bound kernel can be held
    against (it is neither an HBM nor an MFMA kernel)."""
    import numpy as np
    solver.set_stragglers(0)          # every ant finishes inside its own launch here: launch time = the longest walk
    solver.init_pheromone(1.0)
    solver.begin(p, ids[0], ids[1], streams=[wl["stream"]])
    ns = []
    for g in range(generations):
        solver.profile(True, 1)
        solver.run(1)
        pr = solver.profile_read()
        _, lens = solver.ants()
        ns.append(pr["walk"]["ms"] * 1e6 / max(int(lens.max()) - 1, 1))
    solver.profile(False, 1)
    solver.set_stragglers(-1)
    step = float(np.median(ns))
    out = {"ns_per_step_of_the_longest_walk": step, "generations_sampled": generations,
           "source": "walk launch time / max(ant steps) per generation"}
    ev = _walk_loop_evidence()
    out.update(ev)
    if {"instructions_per_step", "cycles_per_instruction", "shader_clock_ghz"} <= set(ev):
        floor = ev["instructions_per_step"] * ev["cycles_per_instruction"] / ev["shader_clock_ghz"]
        out["issue_floor_ns"] = floor
        out["frac_of_issue_floor"] = floor / step if step > 0 else 0.0
        if "vector_memory_per_step" in ev:
            # round 6 (profiles/r06/walk_trims.txt): a vector-memory instruction costs ~6 cycles + one per cache line it names (the step's two record loads ~14
            # each, its two touch loads of the 2-hop ball ~21 each: profiles/r02/issue_rates.txt), a conditional branch that is not taken 13 -- the floor of THIS
            # step structure, against which removing eight plain instructions was measured to gain nothing
            cyc = (ev["instructions_per_step"] - ev["vector_memory_per_step"] - ev["conditional_branches_per_step"]) * ev["cycles_per_instruction"] \
                + 2 * 14 + max(ev["vector_memory_per_step"] - 2, 0) * 21 + 13 * ev["conditional_branches_per_step"]
            out["structure_floor_ns"] = cyc / ev["shader_clock_ghz"]
            out["frac_of_structure_floor"] = out["structure_floor_ns"] / step if step > 0 else 0.0
            out["structure_floor_note"] = "plain slots x 4.31 cycles + line cycles of the four vector-memory instructions + two not-taken branches; what is left is the exposed part of the LDS probe and of the record wait"
    return out


def full_run_extra(solver, params, ids, wl, n, gens=500):
    """BASELINE config 3 at its stated length on the same grid: a fresh identical search, 500 generations end to end
    (exploration, convergence, the converged regime).  Outside `value`; bounded (~25 ms of GPU time)."""
    solver.profile(False, 1)
    solver.init_pheromone(1.0)
    solver.begin(params(gens, wl["rng_seed"]), ids[0], ids[1], streams=[wl["stream"]])
    solver.ctx.sync()
    solver.profile(True, 10)
    t0 = time.perf_counter()
    solver.run(gens)
    solver.sync()
    dt = time.perf_counter() - t0
    pr = solver.profile_read()
    solver.profile(False, 1)
    cost, path, _ = solver.result()
    tr = solver.trace()
    fused_ms = pr["evaporate"]["ms"] / max(pr["evaporate"]["launches"], 1)
    return {"workload": "%d^3, %d generations of the same search (BASELINE config 3 as stated)" % (n, gens), "generations": gens,
            "generations_per_s": gens / dt, "ms_total": dt * 1e3, "best_cost": float(cost), "path_nodes": int(len(path)),
            "generation_of_last_improvement": int((tr["bestL"] != tr["bestL"][-1]).sum()),
            "kernel_ms_per_generation": {k: v["ms"] / max(v["launches"], 1) for k, v in pr.items()},
            "in_loop_frac": 48.0 * n ** 3 / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fused_ms > 0 else None}


def ref_mode_extra(solver, ids, n, ants, gens=200):
    """WA_RNG_REF on the same grid: the reference's own libc rand() stream and std::sort tie order, bit for bit (what the REF goldens pin
    against the reference itself).  Ants walk one after another while the colony explores; converged generations are speculated in
    parallel.  Outside `value`; ~1.5 s."""
    from welding_robot_amd import api
    p = api.default_params(max_iteration=gens, predict=PREDICT * ants / ANTS, fixed_colony=ants, rng_mode=api.RNG_REF)
    solver.srand(12345)
    solver.init_pheromone(1.0)
    solver.begin(p, ids[0], ids[1])
    solver.ctx.sync()
    t0 = time.perf_counter()
    solver.run(gens)
    solver.sync()
    dt = time.perf_counter() - t0
    cost, path, _ = solver.result()
    tr = solver.trace()
    return {"workload": "%d^3 / %d ants, %d generations in WA_RNG_REF mode (srand(12345))" % (n, ants, gens), "generations_per_s": gens / dt, "ms_total": dt * 1e3,
            "ns_per_ant_step": dt * 1e9 / max(int(tr["steps"].sum()), 1), "best_cost": float(cost), "path_nodes": int(len(path))}


def live_traffic(n, ants, timeout_s=90):
    """HBM bytes per launch of the in-loop sweep launch, measured in THIS run: two child passes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, as the guide's HBM section prescribes; 30 generations each),
    summed per kernel, FETCH_SIZE doubled (gfx950 tallies the 128-B requests of a 16-B/lane streaming read at 64 B).  Returns
    (bytes per launch, dispatches, source string) or None when rocprofv3 is missing, this process itself runs under a profiler, or a
    pass fails -- the caller then quotes the committed passes (profiles/pmc_traffic.json)."""
    import csv
    import glob
    import shutil
    import tempfile
    if os.environ.get("WA_BENCH_NO_LIVE_PMC") == "1" or shutil.which("rocprofv3") is None:
        return None
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        return None   # never nest profilers
    tmp = tempfile.mkdtemp(prefix="weld_pmc_")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "30", "--warmup", "2", "--no-cpu", "--no-extras", "--no-roofline-256", "--grid", str(n), "--ants", str(ants)]
            env = dict(os.environ, WA_BENCH_NO_LIVE_PMC="1", TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and row["Kernel_Name"].startswith("void k_evap_rank_mark<false, 6>"):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            got[counter] = (sum(vals) / len(vals) * 1024.0, len(vals))   # the counters are in KB
        return (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0], got["FETCH_SIZE"][1],
                "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run (30 generations each, %d dispatches), FETCH_SIZE doubled "
                "per MI355X_MICROARCH.md (an over-count for the small reads of the rank / mark blocks: upper bound)" % got["FETCH_SIZE"][1])
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def multi_start_run(ctx, grid, ids, n, ants, problems, groups, gens, lazy, warm=5, seed=4242):
    """`problems` independent searches of the same grid (different DEV streams) in the slots of one solver, split into `groups`
    pipelined groups (0: the library's rule; wa_acs_set_pipeline): problem-generations/s over generations warm..gens-1, per-launch
    kernel times (sampled every 10th generation), the in-loop sweep's fraction of the HBM peak, the histories."""
    import numpy as np
    from welding_robot_amd import api
    s = api.AcsSolver(ctx, grid, n_slots=problems, max_colony=ants, lazy=lazy)
    s.set_pipeline(groups)
    p = api.default_params(max_iteration=gens, predict=PREDICT * ants / ANTS, fixed_colony=ants, rng_mode=api.RNG_DEV, seed=seed)
    s.init_pheromone(1.0)
    s.begin(p, [ids[0]] * problems, [ids[1]] * problems, streams=list(range(100, 100 + problems)))
    s.run(warm)                                                       # warm-up generations of the same searches
    s.sync()
    s.profile(True, 10)
    t0 = time.perf_counter()
    s.run(gens - warm)
    t_enq = time.perf_counter() - t0                                  # the host's share: wa_acs_run returns when everything is enqueued
    s.sync()
    dt = time.perf_counter() - t0
    pr = s.profile_read()
    costs, _ = s.results(problems)
    hist = np.stack([np.ascontiguousarray(s.trace(q)["bestL"], np.float32).view(np.uint32) for q in range(problems)])
    steps = np.stack([s.trace(q)["steps"] for q in range(problems)])
    used = s.pipeline_groups()
    s.close()
    per = {k: v["ms"] / max(v["launches"], 1) for k, v in pr.items()}
    fused_ms = per["evaporate"]
    out = {"kind": "lazy" if lazy else "dense", "problems": problems, "groups": used,
           "problem_generations_per_s": problems * (gens - warm) / dt, "ms_per_generation_of_all": dt * 1e3 / (gens - warm),
           "host_enqueue_ms_per_generation_of_all": t_enq * 1e3 / (gens - warm), "kernel_ms_per_launch": per,
           "slots_per_launch": problems / used,
           "in_loop_sweep_frac": None if lazy or fused_ms <= 0 else (problems / used) * 48.0 * n ** 3 / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out, hist, steps, costs


def multi_start_extra(ctx, grid, params, ids, n, ants, problems=8, gens=100):
    """BASELINE config 4's workload on ONE GPU: `problems` independent searches of the same grid (different DEV streams: a multi-start
    batch) in the slots of one solver.  A lone search leaves the chip almost empty (256 wavefronts); eight fill 2 048 wave slots, and their
    sweeps stream 8 x 100 MB per generation -- past the 256 MiB Infinity Cache, so `in_loop_frac` (one launch for all eight: groups = 1)
    is an HBM figure.  `pipelined`: the same batch as the library runs it by default -- two groups of slots on streams of their own, one
    group's sweep under the other's walk (wa_acs_set_pipeline) -- and the same with lazy evaporation; all histories must be equal.
    Outside `value`."""
    import numpy as np
    out = {"workload": "%d independent %d^3 / %d-ant searches on one GPU, generations 5..%d of each" % (problems, n, ants, gens - 1)}
    one, h1, s1, costs = multi_start_run(ctx, grid, ids, n, ants, problems, 1, gens, False)
    pip, h2, s2, _ = multi_start_run(ctx, grid, ids, n, ants, problems, 0, gens, False)
    lz1, h3, s3, _ = multi_start_run(ctx, grid, ids, n, ants, problems, 1, gens, True)
    lzp, h4, s4, _ = multi_start_run(ctx, grid, ids, n, ants, problems, 0, gens, True)
    out["one_stream"] = one
    out["pipelined"] = pip
    out["lazy_one_stream"] = lz1
    out["lazy_pipelined"] = lzp
    out["problem_generations_per_s"] = pip["problem_generations_per_s"]
    out["lazy_problem_generations_per_s"] = lzp["problem_generations_per_s"]
    out["in_loop_frac"] = one["in_loop_sweep_frac"]
    out["bytes_per_launch"] = problems * 48.0 * n ** 3
    out["best_costs"] = [float(c) for c in costs]
    out["pipelined_identical_histories"] = bool(np.array_equal(h1, h2) and np.array_equal(s1, s2))
    out["lazy_identical_histories"] = bool(np.array_equal(h1, h3) and np.array_equal(h1, h4) and np.array_equal(s1, s3) and np.array_equal(s1, s4))
    return out


def multi_start_curve_extra(ctx, grid, ids, n, ants, gens=60):
    """Problems-per-GPU curve: P = 1 .. 32 independent searches per solver, dense and lazy, on one stream and as the library pipelines them
    by rule; problem-generations/s over generations 5..gens-1, per-launch kernel times, in-loop sweep fraction.  Outside `value`; bounded."""
    import numpy as np
    rows = []
    for lazy in (False, True):
        for P in (1, 2, 4, 8, 16, 32):
            ref = None
            for G in (1, 0):
                r, h, st, _ = multi_start_run(ctx, grid, ids, n, ants, P, G, gens, lazy)
                if ref is None:
                    ref = (h, st)
                elif r["groups"] == 1:
                    continue                                          # the rule keeps this batch on one stream: already measured
                r["identical_to_one_stream"] = bool(np.array_equal(h, ref[0]) and np.array_equal(st, ref[1]))
                rows.append(r)
    return {"workload": "%d^3 / %d-ant multi-start batches, generations 5..%d" % (n, ants, gens - 1), "rows": rows}


def c5_full_extra(ctx):
    """BASELINE config 5 at FULL size on this one GPU: 256^3 grid, 64 weld points, all 2 016 pair searches x 150 generations
    (24 ants each, lazy evaporation, concurrent slots sized by the library's rule from free memory), then the 64-seam ACS-TSP order
    from the in-memory matrix.  Solver creation (device allocation) is timed separately.  Outside `value`; bounded (a few s)."""
    import importlib.util
    import numpy as np
    from welding_robot_amd import api, synth
    spec = importlib.util.spec_from_file_location("plan_batch", os.path.join(ROOT, "examples", "plan_batch.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    n, P, gens, seed = 256, 64, 150, 7
    t0 = time.perf_counter()
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    pts = synth.synth_weld_points(free, n, P, seed=seed)
    t_inputs = time.perf_counter() - t0
    predict = float(24 / 0.35)
    waited = pb.wait_for_device_memory(ctx)   # (a process that has just exited may still be giving its memory back, see examples/plan_batch.py)
    GiB = float(1 << 30)
    runs = []
    for rep in range(2):
        # first: the solver is built from what the context kept of the earlier extras' solvers (other shapes: the arena re-maps their chunks)
        # plus fresh memory, which the driver wipes first when an earlier process has used it (~25-40 ms per GB: that, not the library, is the first creation);
        # second: the same job again, as a planning service (or the drop-in, which creates its solver per searchBestPathOfPoints call) pays it
        st0 = ctx.cache_stats()
        t0 = time.perf_counter()
        cost, paths, mine = pb.plan(ctx, grid, pts, gens, predict, seed, 0, lazy=True)
        ctx.sync()
        t_pairs = time.perf_counter() - t0 - pb.plan.last_create_s
        st1 = ctx.cache_stats()
        runs.append({"t_solver_create_s": pb.plan.last_create_s, "t_pairs_s": t_pairs, "slots": pb.plan.last_slots,
                     "batch_solve_reset_read_s": pb.plan.last_batch_s,
                     "cache_hit_gib": (st1["hit_bytes"] - st0["hit_bytes"]) / GiB, "cache_miss_gib": (st1["miss_bytes"] - st0["miss_bytes"]) / GiB,
                     "released_to_driver_gib": (st1["released_bytes"] - st0["released_bytes"]) / GiB, "out_of_memory_events": st1["oom_events"] - st0["oom_events"],
                     "kept_gib_after": st1["kept_bytes"] / GiB})
        if rep == 0:
            first_cost = cost
    same = bool(np.array_equal(first_cost, cost))
    t0 = time.perf_counter()
    tour = api.gtsp_solve(ctx, cost, mode=api.RNG_DEV, seed=seed)
    t_gtsp = time.perf_counter() - t0
    grid.close()
    pairs = P * (P - 1) // 2
    warm = runs[1]
    return {"workload": "256^3 grid, 64 weld points = %d pair searches x %d generations, 24 ants, lazy evaporation; then the 64-seam order" % (pairs, gens),
            "slots_by_rule": warm["slots"], "batches": -(-pairs // warm["slots"]), "t_pairs_s": warm["t_pairs_s"], "t_solver_create_s": warm["t_solver_create_s"],
            "t_pairs_first_s": runs[0]["t_pairs_s"], "t_solver_create_first_s": runs[0]["t_solver_create_s"],
            "cache": {"arena": bool(ctx.cache_stats()["arena"]), "first": runs[0], "second": runs[1],
                      "note": "t_solver_create_s / t_pairs_s: the job run a second time on the same context (every block served from kept memory); *_first_s: the first "
                              "time, behind the other extras' solvers -- cache_miss_gib of it is fresh device memory, which the driver wipes before handing it out if an earlier process on the box has used it"},
            "identical_costs_both_times": same,
            "t_gtsp_s": t_gtsp, "t_host_inputs_s": t_inputs, "t_memory_wait_s": waited, "pair_generations_per_s": pairs * gens / warm["t_pairs_s"],
            "all_reached": bool(np.isfinite(cost).all()), "tour_cost": float(tour["L"][0]), "tour_iterations": int(tour["iters"][0])}


def c5_sharded_extra(ctx, comm, rank, world, check_against_one_rank=True):
    """BASELINE config 5 as the multi-GPU job it is (N > 1, or WA_FORCE_DIST=1): STRONG scaling -- the 2 016 pair searches of ONE planning job
    dealt over the ranks.  Rank 0 builds the 256^3 grid and wa_comm_broadcast_grid ships it (occupancy + axis tables: ncclBroadcast);
    the pairs are dealt longest-processing-time-first in end-point groups (the drop-in's rule: welding_robot_amd/dist.py = ACSRank_3D.hpp drop-in
    `deal`); every rank plans its share (lazy evaporation, slots by rule); wa_comm_allgather_costs brings every cost to every rank and
    wa_comm_gather_paths every path to rank 0, which orders the seams (ACS_GTSP.hpp:224-253, :286-298 need the whole matrix on one rank).
    Reported: aggregate pair-generations/s over the slowest rank's wall time incl. the exchanges, the per-rank search times and their
    imbalance, the exchange times; checked: cost matrix and tour equal the ONE-rank run of the same job (rank 0 repeats it alone).
    WA_BENCH_C5="grid,points,generations" shrinks the job (tests: several ranks on one GPU).  Outside `value`."""
    import importlib.util
    import numpy as np
    from welding_robot_amd import api, synth
    from welding_robot_amd import dist as wd
    spec = importlib.util.spec_from_file_location("plan_batch", os.path.join(ROOT, "examples", "plan_batch.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    n, P, gens = (int(v) for v in os.environ.get("WA_BENCH_C5", "256,64,150").split(","))
    slots = int(os.environ.get("WA_BENCH_C5_SLOTS", "0"))
    seed, predict = 7, float(24 / 0.35)
    free = None
    t0 = time.perf_counter()
    mine_grid, grid_err = None, None
    if rank == 0:   # (a root that cannot build its grid must say so BEFORE the others enter the broadcast: ADVICE r05)
        try:
            free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
            mine_grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
        except Exception as e:   # noqa: BLE001
            grid_err = "rank 0 could not build the grid: %r" % (e,)
            print("[bench] c5_sharded: " + grid_err, file=sys.stderr, flush=True)
    if comm.allreduce([0.0 if grid_err is None else 1.0], "max")[0] != 0.0:   # (doubles as the barrier in front of the timed broadcast)
        return {"error": grid_err or "rank 0 could not build the grid"} if rank == 0 else None
    t1 = time.perf_counter()
    grid = comm.broadcast_grid(mine_grid, root=0)
    t_bcast = time.perf_counter() - t1
    if free is None:
        free = grid.occupancy()                       # (the weld points are drawn from the grid every rank now holds)
    pts = synth.synth_weld_points(free, n, P, seed=seed)
    pair_list = [(i, j) for i in range(P) for j in range(i + 1, P)]
    comm.barrier()
    t2 = time.perf_counter()
    # a rank whose share fails (out of device memory, say) must not leave the others waiting in the exchange: every rank reports, and
    # all of them give the leg up together
    try:
        if os.environ.get("WA_BENCH_FAIL_RANK") == str(rank):     # (tests/test_gpu_mock_ranks.py: one rank fails, nobody hangs)
            raise RuntimeError("WA_BENCH_FAIL_RANK")
        cost, paths, n_mine = pb.plan(ctx, grid, pts, gens, predict, seed, slots, rank, world, lazy=True)
        ctx.sync()
        failed = None
    except Exception as e:   # noqa: BLE001
        failed = "rank %d: %r" % (rank, e)
        print("[bench] c5_sharded: " + failed, file=sys.stderr, flush=True)
    if comm.allreduce([0.0 if failed is None else 1.0], "max")[0] != 0.0:
        comm.barrier()
        grid.close()
        return {"error": failed or "another rank could not plan its share (see its stderr)"} if rank == 0 else None
    t_search = time.perf_counter() - t2 - pb.plan.last_create_s
    t3 = time.perf_counter()
    index_of = {ij: k for k, ij in enumerate(pair_list)}
    mine = sorted(index_of[ij] for ij in paths)
    vec = comm.allgather_costs(mine, [cost[pair_list[k]] for k in mine], len(pair_list))
    full = np.zeros((P, P), np.float64)
    for k, (i, j) in enumerate(pair_list):
        full[i, j] = full[j, i] = vec[k]
    gathered = comm.gather_paths({k: paths[pair_list[k]] for k in mine}, root=0)
    t_exchange = time.perf_counter() - t3
    t_job = time.perf_counter() - t2 - pb.plan.last_create_s       # search + exchange on this rank (solver creation: a one-off, reported beside it)
    t_all = comm.allreduce([t_search, t_job, t_exchange, pb.plan.last_create_s], "max")
    t_sum = comm.allreduce([t_search, float(n_mine)], "sum")
    t_min = comm.allreduce([t_search], "min")
    def rank0_report():
        tour = api.gtsp_solve(ctx, full, mode=api.RNG_DEV, seed=seed)
        pairs = len(pair_list)
        out = {"workload": "%d^3 grid, %d weld points = %d pair searches x %d generations dealt over %d rank(s), lazy evaporation; then the seam order on rank 0"
                           % (n, P, pairs, gens, world),
               "scaling": "strong", "ranks": world, "rccl": comm.stats(), "pairs": pairs, "pairs_per_rank_mean": t_sum[1] / world, "slots_rank0": pb.plan.last_slots,
               "pair_generations_per_s": pairs * gens / t_all[1], "t_job_s_slowest_rank": t_all[1],
               "t_search_s": {"slowest": t_all[0], "fastest": t_min[0], "mean": t_sum[0] / world, "imbalance_max_over_mean": t_all[0] / (t_sum[0] / world)},
               "t_exchange_s_slowest_rank": t_all[2], "t_grid_broadcast_s": t_bcast, "grid_broadcast_bytes": int(n) ** 3 + 12 * n,
               "t_solver_create_s_slowest_rank": t_all[3], "paths_on_rank0": len(gathered), "all_reached": bool(np.isfinite(full).all()),
               "tour_cost": float(tour["L"][0]), "tour_iterations": int(tour["iters"][0])}
        if check_against_one_rank:
            c1, p1, _ = pb.plan(ctx, grid, pts, gens, predict, seed, slots, 0, 1, lazy=True)
            t1r = api.gtsp_solve(ctx, c1, mode=api.RNG_DEV, seed=seed)
            out["equals_one_rank_run"] = bool(np.array_equal(c1, full) and np.array_equal(t1r["edges"][0], tour["edges"][0]) and
                                              all(np.array_equal(p1[pair_list[k]], gathered[k]) for k in gathered) and len(gathered) == pairs)
            if not out["equals_one_rank_run"]:
                print("[bench] c5_sharded: the sharded C5 run DIFFERS from the one-rank run", file=sys.stderr, flush=True)
        return out

    out = None
    if rank == 0:   # (rank 0 alone from here to the barrier: whatever happens, it gets there)
        try:
            out = rank0_report()
        except Exception as e:   # noqa: BLE001
            print("[bench] c5_sharded, rank 0: %r" % (e,), file=sys.stderr, flush=True)
            out = {"error": repr(e)[:500]}
    comm.barrier()
    grid.close()
    return out


def pair_planning_extra(ctx, grid, free, n):
    """Secondary, informational: BASELINE config C5's shape on the bench grid -- all pair searches between 16 weld
    points (120 pairs x 150 generations, 24 ants each, 32 concurrent slots) with the dense sweep and with lazy
    evaporation (wa_acs_create_lazy); the two must agree bit for bit.  Not part of `value`."""
    import importlib.util
    import numpy as np
    from welding_robot_amd import synth
    spec = importlib.util.spec_from_file_location("plan_batch", os.path.join(ROOT, "examples", "plan_batch.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    pts = synth.synth_weld_points(free, n, 16, seed=7)
    gens, predict = 150, float(24 / 0.35)
    res = {}
    for lazy in (False, True):
        pb.plan(ctx, grid, pts[:4], 5, predict, 7, 8, lazy=lazy)      # warm-up (allocation, first launches)
        ctx.sync()
        t0 = time.perf_counter()
        cost, paths, _ = pb.plan(ctx, grid, pts, gens, predict, 7, 32, lazy=lazy)
        ctx.sync()
        res[lazy] = (time.perf_counter() - t0, cost, paths)
    same = bool(np.array_equal(res[False][1], res[True][1]) and all(np.array_equal(res[False][2][k], res[True][2][k]) for k in res[False][2]))
    pairs = len(pts) * (len(pts) - 1) // 2
    return {"workload": "%d^3 grid, 16 weld points = %d pair searches x %d generations, 24 ants, 32 slots" % (n, pairs, gens),
            "dense_sweep_pair_generations_per_s": pairs * gens / res[False][0],
            "lazy_evaporation_pair_generations_per_s": pairs * gens / res[True][0],
            "identical_costs_and_paths": same}


def wait_for_device_memory(ctx, want=0.85, timeout_s=30.0):
    """A process that has just exited may still be giving its device memory back; allocations made meanwhile can end up in
    host-visible memory (measured: the whole run 4x slower).  Wait until most of the device memory is free."""
    t0 = time.perf_counter()
    while True:
        free, total = ctx.memory_info()
        if free >= want * total or time.perf_counter() - t0 > timeout_s:
            return time.perf_counter() - t0
        time.sleep(0.25)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: relaunch under torchrun as a CHILD process (never exec after GPU init)
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    import numpy as np
    import torch
    import torch.distributed as dist

    from welding_robot_amd import api, synth
    from welding_robot_amd import dist as wd

    rank, local_rank, world = wd.env_rank()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # WA_FORCE_DIST=1 runs the RCCL path even with one rank (torchrun --nproc-per-node 1): lets a
    # 1-GPU box exercise exactly the code the 2/4/8-GPU runs execute
    dist_on = world > 1 or (os.environ.get("WA_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (WA_BENCH_BACKEND=gloo: torch's own collectives over gloo on host tensors -- how tests/test_gpu_mock_ranks.py runs this very path with two
        #  ranks on ONE GPU, where RCCL refuses a second rank; the library's exchange then goes through the test's RCCL stand-in)
        if os.environ.get("WA_BENCH_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if dist_on and dist.get_backend() == "gloo" else dev   # where torch's own small collectives live
    ctx = api.Context(local_rank)  # raises if libweldacs.so or the device is missing: no fallback
    mem_wait_s = wait_for_device_memory(ctx)
    n, K, W = args.grid, args.steps, args.warmup
    wl = wd.per_rank_workload(rank if args.workload_index is None else args.workload_index)
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=wl["grid_seed"], occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    solver = api.AcsSolver(ctx, grid, n_slots=1, max_colony=args.ants)

    def params(iters, seed):
        return api.default_params(max_iteration=iters, predict=PREDICT, fixed_colony=args.ants, rng_mode=api.RNG_DEV, seed=seed)

    # ---- warm-up: W generations of a throw-away search (different key), untimed
    if W > 0:
        solver.init_pheromone(1.0)
        solver.begin(params(W, wl["rng_seed"] + 1000), ids[0], ids[1], streams=[wl["stream"]])
        solver.run(W)
        solver.sync()
    # ---- per-problem setup, untimed (reported)
    t_setup = time.perf_counter()
    solver.init_pheromone(1.0)
    solver.begin(params(K, wl["rng_seed"]), ids[0], ids[1], streams=[wl["stream"]])
    ctx.sync()
    setup_ms = (time.perf_counter() - t_setup) * 1e3
    # (stamping a launch costs the stream ~8 us -- measured: every generation stamped = -4 % on `value` -- so only every n-th is)
    solver.profile(True, args.profile_every)
    # global-best exchange (C4): libweldacs.so's own RCCL communicator (wa_comm_*, csrc/host_comm.inc -- what a C++ host uses);
    # torch.distributed only ships the 128-byte id and provides the contract's barrier.  WA_BENCH_TORCH_ALLREDUCE=1 selects the
    # round-2 path (torch.distributed all_reduce of the exported trace) instead.
    chunk = min(CHUNK, K)
    use_torch_ar = dist_on and os.environ.get("WA_BENCH_TORCH_ALLREDUCE") == "1"
    comm = None
    if dist_on and not use_torch_ar:
        uid = torch.from_numpy(api.Comm.unique_id() if rank == 0 else np.zeros(128, np.uint8)).to(cdev)
        dist.broadcast(uid, src=0)
        try:
            comm = api.Comm(ctx, rank, world, uid.cpu().numpy())
            comm.barrier()
            ok = 1.0
        except api.WeldacsError as e:   # (never seen; every rank must take the same path, so the ranks agree on it below)
            print("[bench] rank %d: wa_comm_create failed (%s)" % (rank, e), file=sys.stderr)
            comm, ok = None, 0.0
        if wd.max_over_ranks(-ok, cdev) != -1.0:   # some rank failed: everybody exchanges through torch.distributed instead
            if comm is not None:
                comm.close()
            comm, use_torch_ar = None, True
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev) if use_torch_ar else None
    gbuf = [torch.empty(chunk, dtype=torch.float32, device=dev) for _ in range((K + chunk - 1) // chunk)] if use_torch_ar else []
    works = []

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    dbg16 = np.zeros(16, np.uint64)
    ctx.check(ctx.lib.wa_acs_debug_counters(solver.h, dbg16.ctypes.data, 1))   # (reset: stragglers handed over / resumed in the timed region)
    barrier()
    t0 = time.perf_counter()
    done = 0
    while done < K:
        c = min(chunk, K - done)
        solver.run(c)
        if comm is not None:   # MIN over ranks of best_L[done .. done+c), asynchronous on the communicator's stream
            comm.allreduce_best(solver, done, c)
        elif use_torch_ar:
            gb = gbuf[done // chunk]
            solver.export_trace(gb.data_ptr(), done, c)
            with torch.cuda.stream(ext):
                works.append(wd.allreduce_min_(gb[:c], async_op=True))
        done += c
    solver.sync()
    glob = comm.read_best(0, K) if comm is not None else None   # waits for the exchanges
    owner = [int(v[0]) for v in comm.read_best_owner(K - 1, 1)[1:]] if comm is not None else None   # (rank, slot) that holds the final global best's path
    for w in works:
        if w is not None:
            w.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = wd.max_over_ranks(elapsed, cdev)
    total_gens = wd.sum_over_ranks(K, cdev)
    ctx.check(ctx.lib.wa_acs_debug_counters(solver.h, dbg16.ctypes.data, 0))
    stragglers = {"handed_over": int(dbg16[9]), "finished_by_resume_blocks": int(dbg16[7]),
                  "note": "ants that could no longer be among the depositing ranks left their walk launch at one of the loop's checks and were finished "
                          "beside the next generation's ants on the previous generation's field (DESIGN 4e): every step taken, the two counts are equal"}

    prof = solver.profile_read()
    cost, path, _ = solver.result()
    trace = solver.trace()
    if dist_on:  # the reduced global-best history must be the element-wise MIN of the ranks' histories
        if glob is None:
            glob = torch.cat([g[:min(chunk, K - i * chunk)] for i, g in enumerate(gbuf)]).cpu().numpy()
        assert glob.shape[0] == K and np.all(glob <= trace["bestL"] + 0.0), "global best exceeds a local best"
        if world == 1:
            assert np.array_equal(glob.view(np.uint32), trace["bestL"].view(np.uint32))
        else:   # ... and somebody's local best: the MIN over ranks of the local histories, computed the slow way
            allh = [torch.empty(K, dtype=torch.float32, device=cdev) for _ in range(world)]
            dist.all_gather(allh, torch.from_numpy(np.ascontiguousarray(trace["bestL"], np.float32)).to(cdev))
            assert np.array_equal(torch.stack(allh).min(0).values.cpu().numpy().view(np.uint32), glob.view(np.uint32)), "RCCL MIN != MIN of the gathered histories"
    best_all = wd.max_over_ranks(-float(cost), cdev) * -1.0  # min over ranks
    if rank == 0:
        # ---- the roofline kernel = the launch of the TIMED loop that carries the evaporation sweep (k_evap_rank_mark: 4096 sweep
        # blocks + the rank / mark blocks), per-dispatch HIP events on the library's stream over the timed region
        fused = dict(prof["evaporate"])
        fused_where = "%d launches of the timed region" % fused["launches"]
        if fused["launches"] < 32 and not args.no_extras:   # a short timed region: the same search goes on, untimed, every launch stamped, until 32 samples exist
            solver.profile(True, 1)
            solver.run(32 - int(fused["launches"]))
            more = solver.profile_read()["evaporate"]
            fused = {"ms": fused["ms"] + more["ms"], "launches": fused["launches"] + more["launches"]}
            fused_where += " + %d of the same search continued behind it (stamping every launch inside the timed region would cost it 4 %%)" % more["launches"]
        fused_ms = fused["ms"] / max(fused["launches"], 1)
        # ---- the same launch with the measurement's own share taken out: a dispatch WITH events directly behind one WITHOUT (the walk)
        # starts its workgroups 1.2 us slower and reports 1.6-2 us more than the same dispatch behind another event-carrying one -- blocks,
        # clock, L2 and TLB counters unchanged (profiles/r06/sweep_gap.txt).  `paired`: a stamped no-op dispatch in front of each timed launch
        paired_ms = None
        if not args.no_extras:
            solver.profile(True, 1, paired=True)
            solver.run(32)
            pp = solver.profile_read()["evaporate"]
            paired_ms = pp["ms"] / max(pp["launches"], 1)
            solver.profile(False, 1)
        # ---- beside it: the sweep kernel on its own, at the BASELINE size and past the Infinity Cache
        r128 = sweep_roofline(ctx, n)
        r256 = sweep_roofline(ctx, 256) if not args.no_roofline_256 else None
        alg_bytes = r128["algorithmic_bytes_per_launch"]  # 6 fp32 read + written per voxel
        achieved = alg_bytes / (fused_ms * 1e-3) / 1e9 if fused_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (bench.py cannot host rocprofv3
        # itself): only quoted for the kernel and size they were measured on
        traffic, traffic_alone, traffic_256, traffic_src = None, None, None, None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            g = pt["kernels"]["k_evaporate"]["grids"]
            if str(n) in g:
                traffic_alone = float(g[str(n)]["fetch_bytes_corrected"] + g[str(n)]["write_bytes"])
            if "256" in g and r256:
                traffic_256 = float(g["256"]["fetch_bytes_corrected"] + g["256"]["write_bytes"])
            g = pt["kernels"].get("k_evap_rank_mark", {}).get("grids", {})
            if str(n) in g:
                traffic = float(g[str(n)]["fetch_bytes_corrected"] + g[str(n)]["write_bytes"])
            traffic_src = "committed: " + pt["source"]
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_extras:   # measured in this run when rocprofv3 can wrap a child pass
            lt = live_traffic(n, args.ants)
            if lt is not None:
                traffic, traffic_src = lt[0], lt[2] + "; sweep_alone.traffic_*: " + (traffic_src or "n/a")
        out = {
            "metric": "acs_generations_per_sec", "value": wd.aggregate_rate(total_gens, elapsed), "unit": "generations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3 per GPU: %d^3 synthetic random-obstacle grid (10%% occupied, splitmix64 seed 2024+rank), "
                                   "%d ants fixed, %d generations, alpha 1 beta 0.6 rho 0.8, DEV rng seed 12345+rank; "
                                   "one independent problem per GPU (C4)" % (n, args.ants, K),
                       "grid": [n, n, n], "ants": args.ants, "generations": K, "problems_per_gpu": 1,
                       "global_best_allreduce": ("RCCL MIN over ranks per generation, chunks of %d, async, %s" % (
                           chunk, "torch.distributed" if use_torch_ar else "libweldacs wa_acs_allreduce_best (ncclAllReduce ncclMin on the communicator's stream)"))
                       if dist_on else "n/a (1 GPU)"},
            "roofline": {"bound": "hbm",
                         "kernel": "k_evap_rank_mark<false,6> -- the launch of the timed loop that carries the evaporation sweep (ACSRank_3D.hpp:268-272) "
                                   "beside the rank / mark blocks; per-dispatch HIP events on the library's stream, " + fused_where,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": fused_ms, "sampled_launches": fused["launches"],
                         "paired": None if not paired_ms else {
                             "avg_launch_ms": paired_ms, "frac": alg_bytes / (paired_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "sampled_launches": 32,
                             "note": "the same launch of the same search, 32 more generations, each timed launch preceded by a no-op dispatch that carries events "
                                     "too: `frac` above includes 1.6-2 us that a dispatch with events reports when its predecessor (the walk) has none -- "
                                     "slower workgroup start, same blocks, clock and cache counters (profiles/r06/sweep_gap.txt); `frac` stays the figure that "
                                     "agrees with the rocprofv3 kernel stats, whose start stamp of such a dispatch coincides with its predecessor's end"},
                         "frac_hbm": (r256["achieved"] / HBM_PEAK_GBS) if r256 else None,
                         "note": "at 128^3 both 48 MiB buffers sit in the 256 MiB Infinity Cache: frac is an on-die figure; frac_hbm (= sweep_alone.frac_256: the same sweep "
                                 "on a 256^3 field, 805 MB per launch, non-temporal loads and stores by the library's rule) is the HBM one; multi_start.in_loop_frac is the "
                                 "in-loop HBM figure (eight 128^3 fields per launch)",
                         "sweep_alone": {"kernel": "k_evaporate, launched alone %d times after the timed region" % r128["launches"],
                                         "frac_128": r128["achieved"] / HBM_PEAK_GBS, "avg_launch_ms_128": r128["avg_launch_ms"], "traffic_128": traffic_alone,
                                         "frac_256": (r256["achieved"] / HBM_PEAK_GBS) if r256 else None, "traffic_256": traffic_256, "sweep_256": r256},
                         "end_to_end_frac": alg_bytes * (K / elapsed) / 1e9 / HBM_PEAK_GBS},
            "kernel_ms_sampled": dict({k: v["ms"] / max(v["launches"], 1) for k, v in prof.items()},
                                      generations=[g for g in range(K) if g % args.profile_every == 0],
                                      note="average over the STAMPED generations only (every %d-th of the timed region: hipEventRecord around each launch, which "
                                           "costs the stream ~8 us per stamp, so the sum exceeds ms_per_step); the unperturbed per-kernel times of this command are "
                                           "the rocprofv3 kernel stats under profiles/" % args.profile_every),
            "global_best_owner": owner, "stragglers": stragglers, "setup_ms": setup_ms, "waited_for_device_memory_s": mem_wait_s, "best_cost": float(cost), "best_cost_all_ranks": best_all, "path_nodes": int(len(path)),
            "steps_per_generation_first_last": [int(trace["steps"][0]), int(trace["steps"][-1])],
            "device": ctx.device_name,
        }
        if world == 1 and not args.no_extras:
            # the extras stand outside `value`: one that fails is reported under its key and must not cost the line
            def guarded(fn, *a):
                try:
                    if os.environ.get("WA_BENCH_FAIL") in ("all", fn.__name__):    # (tests/test_gpu_bench.py: the line survives a failing extra)
                        raise RuntimeError("WA_BENCH_FAIL")
                    return fn(*a)
                except Exception as e:   # noqa: BLE001
                    print("[bench] %s failed: %r" % (fn.__name__, e), file=sys.stderr, flush=True)
                    return {"error": repr(e)[:500]}
            out["walk_step"] = guarded(walk_step_extra, solver, params(K, wl["rng_seed"]), ids, wl)
            out["full_run"] = guarded(full_run_extra, solver, params, ids, wl, n)
            out["ref_mode"] = guarded(ref_mode_extra, solver, ids, n, args.ants)
            out["multi_start"] = guarded(multi_start_extra, ctx, grid, params, ids, n, args.ants)
            out["multi_start_curve"] = guarded(multi_start_curve_extra, ctx, grid, ids, n, args.ants)
            out["c5_pair_planning"] = guarded(pair_planning_extra, ctx, grid, free, n)
            solver.close()
            out["c5_full"] = guarded(c5_full_extra, ctx)
    c5s = None
    if comm is not None and (not args.no_extras or os.environ.get("WA_BENCH_C5")):
        # every rank takes part (rank 0 reports): BASELINE config 5 as ONE job over the ranks -- the leg where N GPUs shorten a job (strong
        # scaling); the headline above is C4, one independent search per GPU (weak)
        solver.close() if world > 1 else None
        try:
            c5s = c5_sharded_extra(ctx, comm, rank, world)
        except Exception as e:   # noqa: BLE001  (a failed exchange is reported by every rank together: csrc/host_comm.inc)
            print("[bench] rank %d: c5_sharded failed: %r" % (rank, e), file=sys.stderr, flush=True)
            c5s = {"error": repr(e)[:500]}
    if rank == 0:
        if comm is not None:
            # what RCCL ITSELF says about the communicator the line was produced on (not WORLD_SIZE): ranks = ncclCommCount, version = ncclGetVersion
            # (0: the one-GPU stand-in of tests/mock_rccl answered), collectives issued by rank 0 (VERDICT r05 task 6c)
            try:
                out["rccl"] = comm.stats()
            except Exception as e:   # noqa: BLE001
                out["rccl"] = {"error": repr(e)[:200]}
        if c5s is not None:
            out["c5_sharded"] = c5s
            out["c5_sharded_ranks"] = c5s.get("ranks") if isinstance(c5s, dict) else None
        if world == 1 and not args.no_cpu:
            # the CPU leg comes last (the extras before it are host-paced: 0.52-0.84 s for the C5 extra from box to box, whatever runs in front)
            out.update(cpu_baseline(args, free, n, trace, wl, path, K, elapsed * 1e3))
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
