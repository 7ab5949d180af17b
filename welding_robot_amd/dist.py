"""Multi-GPU sharding of independent planning problems.

The path shards by PROBLEM: weld-point pairs / multi-start replicas / independent grids share
nothing but the read-only occupancy, so every rank owns whole problems and there is no
data-path collective.  The only exchange is the global-best path cost per generation: each rank
exports best_L[g] for a chunk of generations and one MIN all-reduce (RCCL over xGMI on GPUs,
gloo in the CPU tests) of that small vector publishes the global best of every generation in
the chunk, overlapped with the next chunk's walk.  The global best never feeds back into a
problem's own colony/Q parameters, so per-problem results equal the single-GPU run."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_problems(n_problems, rank, world):
    """round-robin problem index -> rank: for problems of equal cost (C4's one search per GPU, multi-start replicas)"""
    return list(range(rank, n_problems, world))


def deal_pairs(pairs, weights, world):
    """Pair searches -> ranks, longest-processing-time-first (the rule of the drop-in ACS_Rank, ACSRank_3D.hpp drop-in
    searchBestPathOfPoints): a search costs about as much as its walks are long, i.e. grows with the Manhattan distance of
    its two points.  Units sorted by weight, each to the least loaded rank; a unit is an END-POINT GROUP (all pairs (i, j)
    with the same j share one heuristic field on the device) while there are >= 4 groups per rank, a single pair otherwise.
    Within a rank: groups side by side, the group with the longest search first and the longest search first inside a group
    (the partly filled last batch holds the shortest searches).  Returns (per-rank lists of pair indices, per-rank load)."""
    n_points = 1 + max((j for _, j in pairs), default=0)
    if world > 1 and n_points - 1 >= 4 * world:
        units = [[k for k, (_, j) in enumerate(pairs) if j == e] for e in range(n_points)]
    else:
        units = [[k] for k in range(len(pairs))]
    order = sorted(((-sum(weights[k] for k in u), i) for i, u in enumerate(units) if u))
    shards, load = [[] for _ in range(world)], [0] * world
    for negw, i in order:
        d = min(range(world), key=lambda r: (load[r], r))
        shards[d] += units[i]
        load[d] -= negw
    for sh in shards:
        gmax = {}
        for k in sh:
            gmax[pairs[k][1]] = max(gmax.get(pairs[k][1], 0), weights[k])
        sh.sort(key=lambda k: (-gmax[pairs[k][1]], pairs[k][1], -weights[k], k))
    return shards, load


def ship_unique_id(rank, world, make_id, port=None, addr=None, timeout_s=120.0):
    """The 128-byte id of a wa_comm from rank 0 to the other ranks of one job without torch / MPI: rank 0 listens on
    MASTER_ADDR : MASTER_PORT + 1000 and hands the bytes to world - 1 connections.  make_id() is only called on rank 0."""
    import socket
    import time
    if world == 1:
        return bytes(bytearray(make_id()))
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or int(os.environ.get("MASTER_PORT", "29500")) + 1000
    if rank == 0:
        uid = bytes(bytearray(make_id()))
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout_s)
        for _ in range(world - 1):
            c, _a = srv.accept()
            c.sendall(uid)
            c.close()
        srv.close()
        return uid
    t0 = time.time()
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() - t0 > timeout_s:
                raise
            time.sleep(0.05)
    buf = b""
    while len(buf) < 128:
        chunk = c.recv(128 - len(buf))
        if not chunk:
            raise RuntimeError("ship_unique_id: rank 0 closed the connection early")
        buf += chunk
    c.close()
    return buf


def per_rank_workload(rank, grid_seed=2024, rng_seed=12345):
    """C4: one independent 128^3 grid per GPU: grid seed 2024 + rank, colony seed 12345 + rank"""
    return dict(grid_seed=grid_seed + rank, rng_seed=rng_seed + rank, stream=rank)


def allreduce_min_(t, async_op=False):
    """global best cost per generation: element-wise MIN over ranks, in place"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    # a process group of ONE rank (torchrun --nproc-per-node 1 with WA_FORCE_DIST=1) still issues the collective:
    # that is how a 1-GPU box runs the RCCL call itself
    return dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=async_op)


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_rate(units_all_ranks, seconds_max):
    """whole-job throughput: units processed by ALL ranks / slowest rank's time"""
    return units_all_ranks / seconds_max if seconds_max > 0 else 0.0
