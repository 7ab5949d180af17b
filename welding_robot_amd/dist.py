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


def order_batch(idx, weights):
    """The searches of ONE batch in the order their slots should have: the blocks of a walk launch start in slot order and the launch lasts as long as
    its longest ant, so the longest searches go first (longest-processing-time-first); the library runs the two halves of the slots as two pipelined
    groups (wa_acs_run: group k = slots [P k / 2, P (k + 1) / 2)), so the sorted list is dealt alternately to the halves.  Measured on BASELINE
    config C5 (224 searches per batch on 256^3): 45.2 -> 43.4 ms per batch (profiles/r06/c5_walk_counters.txt).  Results do not depend on the order:
    every search draws from the stream of its global pair index."""
    srt = sorted(idx, key=lambda k: (-weights[k], k))
    return srt[1::2] + srt[0::2]


_ID_MAGIC = b"WAID1"


def ship_unique_id(rank, world, make_id, port=None, addr=None, timeout_s=120.0, job=None):
    """The 128-byte id of a wa_comm from rank 0 to the other ranks of one job without torch / MPI: rank 0 listens on
    MASTER_ADDR : MASTER_PORT + 1000 and serves every rank 1 .. world-1 until each has ACKNOWLEDGED the id.  A client introduces itself
    with a magic word, its rank, the world size and a job tag (`job`, default: TORCHELASTIC_RUN_ID or MASTER_PORT); anything else that
    connects -- a port probe, a leftover rank of another job -- is turned away and does not take a real rank's place.  A rank counts as
    served when its one-byte acknowledgement has arrived, not when the id was sent: a client that was cut off before it had read the id
    asks again and is served again (same rank, same tag) instead of being turned away while rank 0 goes on as if all was well.
    Every socket has a timeout: a rank that never shows up makes rank 0 fail after timeout_s instead of hanging, and a client that
    was turned away or cut off raises.  make_id() is only called on rank 0."""
    import socket
    import struct
    import time
    if world == 1:
        return bytes(bytearray(make_id()))
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or int(os.environ.get("MASTER_PORT", "29500")) + 1000
    tag = (job or os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT", "29500")).encode()[:32].ljust(32, b"\0")
    hello_len = len(_ID_MAGIC) + 8 + 32

    def recv_exact(c, n):
        buf = b""
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                return None
            buf += chunk
        return buf

    if rank == 0:
        uid = bytes(bytearray(make_id()))
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(4 * world)
        waiting = set(range(1, world))
        t0 = time.time()
        try:
            while waiting:
                left = timeout_s - (time.time() - t0)
                if left <= 0:
                    raise TimeoutError("ship_unique_id: ranks %s never asked for the id" % sorted(waiting))
                srv.settimeout(left)
                try:
                    c, _a = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(5.0)
                    hello = recv_exact(c, hello_len)
                    ok = hello is not None and hello[:len(_ID_MAGIC)] == _ID_MAGIC and hello[len(_ID_MAGIC) + 8:] == tag
                    if ok:
                        r, w = struct.unpack("<ii", hello[len(_ID_MAGIC):len(_ID_MAGIC) + 8])
                        ok = w == world and 1 <= r < world       # (a rank that was served before may ask again: its ack may have been lost)
                    if ok:
                        c.sendall(b"OK" + uid)
                        if recv_exact(c, 1) == b"A":               # the client has the whole id
                            waiting.discard(r)
                    else:
                        c.sendall(b"NO")
                except OSError:
                    pass                      # a stray or broken connection: the rank it claimed to be (if any) stays in `waiting`
                finally:
                    c.close()
        finally:
            srv.close()
        return uid
    t0 = time.time()
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
        except OSError:
            if time.time() - t0 > timeout_s:
                raise
            time.sleep(0.05)
            continue
        try:
            c.settimeout(10.0)
            c.sendall(_ID_MAGIC + struct.pack("<ii", rank, world) + tag)
            head = recv_exact(c, 2)
            if head == b"OK":
                buf = recv_exact(c, 128)
                if buf is not None:
                    c.sendall(b"A")           # only now does rank 0 stop waiting for this rank
                    return buf
            elif head == b"NO":
                raise RuntimeError("ship_unique_id: rank 0 turned rank %d away (another job on this port, or a rank number outside its world)" % rank)
        except OSError:
            pass                              # cut off mid-way (rank 0 not serving yet / a stale listener): try again until the deadline
        finally:
            c.close()
        if time.time() - t0 > timeout_s:
            raise TimeoutError("ship_unique_id: no id from rank 0 within %.0f s" % timeout_s)
        time.sleep(0.05)


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
