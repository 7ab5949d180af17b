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
    """round-robin problem index -> rank (pairs have similar cost; LPT is a later refinement)"""
    return list(range(rank, n_problems, world))


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
