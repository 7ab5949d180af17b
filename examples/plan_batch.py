#!/usr/bin/env python3
"""BASELINE config C5 in miniature or at full size: P weld points on an n^3 grid, all P(P-1)/2
pair searches (ACS_Rank::searchBestPathOfPoints' loop, ACSRank_3D.hpp:472-499) batched across the
slots of each GPU and dealt longest-first across ranks, then the weld-seam order by the ACS-TSP
kernel (ACS_GTSP.hpp:255-284) from the in-memory cost matrix (no graph.in round trip), then on rank 0
the tour's segments stitched and smoothed on the device (main.cpp:283-352).

    python examples/plan_batch.py --grid 256 --points 64 --generations 150          # 1 GPU
    torchrun --nproc-per-node 8 examples/plan_batch.py --grid 256 --points 64       # 8 GPUs (torchrun only starts the processes)

Every pair uses its GLOBAL pair index as DEV-mode stream key, so the cost matrix -- and the tour --
do not depend on the number of ranks or slots.  One process per GPU; the exchanges at the end are the library's own
(RCCL behind the C ABI, no torch): wa_comm_broadcast_grid ships rank 0's grid to the others, wa_comm_allgather_costs brings every pair's cost to every rank (8 KB at P = 64) and
wa_comm_gather_paths every pair's path to rank 0, which orders the seams and stitches (ACS_GTSP.hpp:286-298 needs all of
best_matrix on one rank).  WA_FORCE_DIST=1 runs the same exchanges with the one rank a 1-GPU box has.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from welding_robot_amd import api, synth  # noqa: E402
from welding_robot_amd import dist as wd  # noqa: E402


def plan(ctx, grid, point_ids, generations, predict, seed, slots, rank=0, world=1, fixed_colony=0, lazy=False, neighbourhood=6):
    P = len(point_ids)
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    # dealing and order: longest-processing-time-first over end-point groups (welding_robot_amd/dist.py: deal_pairs -- the rule
    # of the drop-in C++ pair loop).  Results do not depend on either: every search draws from the stream of its global pair index.
    nx, nxy = grid.nx, grid.nx * grid.ny
    vox = [(int(v) // nxy, (int(v) // nx) % grid.ny, int(v) % nx) for v in point_ids]
    weights = [1 + sum(abs(a - b) for a, b in zip(vox[i], vox[j])) for i, j in pairs]
    shards, _ = wd.deal_pairs(pairs, weights, world)
    mine = shards[rank]
    colony = fixed_colony or max(1, int(0.35 * predict / float(grid.precision)))
    by_length = os.environ.get("WA_PLAN_ORDER", "") == "length"
    if by_length:   # experiment (round 6): batches of searches of similar length -- a batch-generation lasts as long as its longest walk
        mine = sorted(mine, key=lambda k: (-weights[k], pairs[k][1], k))
    if not slots:   # sized by rule: free memory, footprint limit, whole batches
        slots, _ = api.pair_slots_by_rule(ctx, grid, colony, max(1, len(mine)), len({pairs[k][1] for k in mine}), generations, lazy=lazy, neighbourhood=neighbourhood,
                                          all_fields=by_length)
    plan.last_slots = slots
    t_create = time.perf_counter()
    solver = api.AcsSolver(ctx, grid, n_slots=slots, max_colony=colony, lazy=lazy, neighbourhood=neighbourhood)
    ctx.sync()
    plan.last_create_s = time.perf_counter() - t_create   # device allocation + field initialisation (one-off for a service)
    p = api.default_params(max_iteration=generations, predict=predict, fixed_colony=fixed_colony,
                           rng_mode=api.RNG_DEV, seed=seed)
    cost = np.zeros((P, P), np.float64)
    paths = {}
    plan.last_batch_s = []                             # seconds per batch: solve / reset / read-back (tools/walk_direct_ab.py --batches)
    for b0 in range(0, len(mine), slots):
        idx = mine[b0:b0 + slots] if by_length else wd.order_batch(mine[b0:b0 + slots], weights)   # (longest searches first in each half of the slots)
        tb = [time.perf_counter()]
        solver.solve(p, [point_ids[pairs[k][0]] for k in idx], [point_ids[pairs[k][1]] for k in idx], streams=idx)
        tb.append(time.perf_counter())
        solver.reset_pheromone(1.0)  # reset() between problems (:481)
        tb.append(time.perf_counter())
        costs, ids_all = solver.results(len(idx))      # one round trip for the whole batch
        tb.append(time.perf_counter())
        plan.last_batch_s.append([round(tb[i + 1] - tb[i], 4) for i in range(3)])
        for q, k in enumerate(idx):
            i, j = pairs[k]
            cost[i, j] = cost[j, i] = costs[q]
            paths[(i, j)] = ids_all[q]
    solver.close()
    return cost, paths, len(mine)


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=96)
    ap.add_argument("--points", type=int, default=16)
    ap.add_argument("--generations", type=int, default=150)
    ap.add_argument("--slots", type=int, default=0, help="concurrent pair searches per GPU; 0 = sized by rule (memory, whole batches)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--neighbourhood", type=int, default=6, choices=(6, 26), help="26: the variant the reference stubs out (ACSRank_3D.hpp:361-388)")
    ap.add_argument("--lazy", action="store_true",
                    help="wa_acs_create_lazy: never-deposited voxels are not swept (same results, O(deposited voxels) per generation)")
    args = ap.parse_args()
    rank, local_rank, world = wd.env_rank()
    ctx = api.Context(local_rank)
    comm = None
    if world > 1 or os.environ.get("WA_FORCE_DIST") == "1":
        uid = wd.ship_unique_id(rank, world, api.Comm.unique_id)      # 128 bytes from rank 0 over a socket: no torch, no MPI
        comm = api.Comm(ctx, rank, world, np.frombuffer(uid, np.uint8))
        comm.barrier()
    wait_for_device_memory(ctx)
    n = args.grid
    if comm is None or rank == 0:
        free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
        grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    else:
        grid = None
    if comm is not None:
        # ONE rank builds the grid (with a mesh that is the O(triangles x voxels) voxelisation, model_grid_map.hpp:223-268), the others
        # receive a replica: occupancy + axis tables by ncclBroadcast (wa_comm_broadcast_grid)
        grid = comm.broadcast_grid(grid, root=0)
        if rank != 0:
            free = grid.occupancy()
    pts = synth.synth_weld_points(free, n, args.points, seed=args.seed)
    predict = float(0.35 ** -1 * 24)  # 24 ants per search at precision 1 (ACSRank_3D.hpp:247)
    t0 = time.perf_counter()
    cost, paths, n_mine = plan(ctx, grid, pts, args.generations, predict, args.seed, args.slots, rank, world, lazy=args.lazy, neighbourhood=args.neighbourhood)
    if comm is not None:
        # every pair is owned by exactly one rank: its cost goes to every rank, its path to rank 0 (the library's own collectives)
        P = args.points
        pair_list = [(i, j) for i in range(P) for j in range(i + 1, P)]
        index_of = {ij: k for k, ij in enumerate(pair_list)}
        mine = sorted(index_of[ij] for ij in paths)
        vec = comm.allgather_costs(mine, [cost[pair_list[k]] for k in mine], len(pair_list))
        cost = np.zeros((P, P), np.float64)
        for k, (i, j) in enumerate(pair_list):
            cost[i, j] = cost[j, i] = vec[k]
        gathered = comm.gather_paths({k: paths[pair_list[k]] for k in mine}, root=0)
        paths = {pair_list[k]: ids for k, ids in gathered.items()}    # rank 0: all of them; elsewhere empty
    t_pairs = time.perf_counter() - t0
    finite = np.isfinite(cost).all()
    t_pairs -= plan.last_create_s
    out = dict(grid=n, points=args.points, neighbourhood=args.neighbourhood, slots=plan.last_slots, lazy_evaporation=bool(args.lazy), t_solver_create_s=plan.last_create_s, pairs=args.points * (args.points - 1) // 2, world=world,
               pairs_this_rank=n_mine, t_pairs_s=t_pairs, all_reached=bool(finite))
    if rank == 0 and finite:
        t1 = time.perf_counter()
        tour = api.gtsp_solve(ctx, cost, mode=api.RNG_DEV, seed=args.seed)
        out.update(tour_cost=float(tour["L"][0]), tour_iterations=int(tour["iters"][0]),
                   order=[int(e[0]) for e in tour["edges"][0]], t_gtsp_s=time.perf_counter() - t1,
                   pair_generations_per_s=out["pairs"] * args.generations / t_pairs)
        # main.cpp:283-352: stitch the tour's segments (rank 0 holds every path), then the two smoothing passes, all on the device
        t2 = time.perf_counter()
        edges = tour["edges"][0][:-1]
        segs = [paths[(min(a, b), max(a, b))] for a, b in edges]
        rev = [1 if a > b else 0 for a, b in edges]          # stored i<j; walk them in tour direction
        path = api.Trajectory.stitch(grid, segs, rev)
        ends = path.points()[[0, -1]]
        s1 = api.Bspline(ctx, 3, 0, 0, 0, len(path))          # BS_Basic<float,3,0,0,0>: time-indexed resampling
        s1.set_param(ends[0], ends[1], path, 150.0)
        n1 = max(16, len(path) // 8)
        _, _, coarse = s1.sample(150.0 / n1, 150.0 / n1, n1, host=False, device=True)
        s2 = api.Bspline(ctx, 3, 3, 2, 2, len(coarse))        # cubic with zero end velocity / acceleration
        z = np.zeros((2, 3), np.float32)
        s2.set_param(np.vstack([ends[:1], z]), np.vstack([ends[1:], z]), coarse, 6000.0)
        traj, ok = s2.sample(0.0, 1.0, 6001)                  # 1 kHz over 6 s
        out.update(stitched_nodes=len(path), coarse_points=len(coarse), trajectory_samples=int(ok.sum()),
                   trajectory_length=float(np.linalg.norm(np.diff(traj, axis=0), axis=1).sum()),
                   t_trajectory_s=time.perf_counter() - t2)
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
