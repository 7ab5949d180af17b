"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Covers what bench.py
and a multi-GPU pair batch do besides the (rank-local) HIP work: problem sharding with no overlap
and no gap, per-rank workloads, the chunked MIN all-reduce of per-generation best costs (the only
collective of the path), max-over-ranks timing and the whole-job rate."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from welding_robot_amd import dist as wd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from welding_robot_amd import dist as wd, api
    rank, local_rank, world = wd.env_rank()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    K, chunk = 130, 50
    rs = np.random.RandomState(100 + rank)
    trace = np.minimum.accumulate(rs.uniform(300, 900, K)).astype(np.float32)   # a monotone best-cost history
    glob = []
    works = []
    for g0 in range(0, K, chunk):                      # same chunking as bench.py
        t = torch.from_numpy(trace[g0:g0 + chunk].copy())
        works.append((t, wd.allreduce_min_(t, async_op=True)))
    for t, w in works:
        w.wait()
        glob.append(t.numpy())
    glob = np.concatenate(glob)
    # the owner-carrying exchange (csrc/host_comm.inc: one MIN all-reduce of the packed key): keys packed by the library's own helper,
    # reduced over gloo as int64 (non-negative float bits keep the top bit clear), unpacked again
    slots = 3
    local = np.stack([np.roll(trace, q) for q in range(slots)])          # this rank's searches
    local[:, :5] = np.float32(777.0)                                     # ties across ranks AND slots in the first generations
    keys = torch.tensor([min(api.pack_best_key(float(local[q, g]), rank, q) for q in range(slots)) for g in range(K)], dtype=torch.int64)
    dist.all_reduce(keys, op=dist.ReduceOp.MIN)
    owners = [api.unpack_best_key(int(k)) for k in keys.tolist()]
    elapsed = 0.010 * (rank + 1)
    out = dict(rank=rank, world=world, mine=trace.tolist(), glob=glob.tolist(), local=local.tolist(),
               owners=[[float(c), r, q] for c, r, q in owners],
               tmax=wd.max_over_ranks(elapsed, dev), total=wd.sum_over_ranks(K, dev),
               shard=wd.shard_problems(2016, rank, world), wl=wd.per_rank_workload(rank))
    json.dump(out, open(sys.argv[1] + ".%%d" %% rank, "w"))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_allreduce_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path / "out")], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(str(tmp_path / "out") + ".%d" % r)) for r in range(2)]
    want = np.minimum(np.array(res[0]["mine"], np.float32), np.array(res[1]["mine"], np.float32))
    for r in res:
        assert np.array_equal(np.array(r["glob"], np.float32), want)  # global best per generation = MIN over ranks
        assert r["tmax"] == 0.020 and r["total"] == 260 and r["world"] == 2
    # packed keys: every rank sees the same (cost, owner rank, owner slot) per generation = the lexicographic minimum over all searches
    allloc = np.stack([np.array(r["local"], np.float32) for r in res])          # [rank][slot][generation]
    for g in range(allloc.shape[2]):
        cands = sorted((float(allloc[r, q, g]), r, q) for r in range(2) for q in range(allloc.shape[1]))
        assert res[0]["owners"][g] == res[1]["owners"][g] == list(cands[0]), g
    assert res[0]["owners"][0] == [777.0, 0, 0]                                 # a tie goes to the lowest rank, then the lowest slot
    assert wd.aggregate_rate(res[0]["total"], res[0]["tmax"]) == 260 / 0.020
    a, b = set(res[0]["shard"]), set(res[1]["shard"])
    assert not (a & b) and a | b == set(range(2016)) and abs(len(a) - len(b)) <= 1
    assert res[0]["wl"] != res[1]["wl"] and res[1]["wl"]["grid_seed"] == 2025 and res[1]["wl"]["rng_seed"] == 12346


def test_comm_id_is_shipped_over_a_socket_without_torch(tmp_path):
    """wd.ship_unique_id: rank 0 hands the 128 bytes of a wa_comm id to the other ranks (examples/plan_batch.py under torchrun);
    three processes, the late ones retry until rank 0 listens"""
    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from welding_robot_amd import dist as wd
        rank, world, port = int(sys.argv[1]), 3, int(sys.argv[2])
        if rank == 0:
            time.sleep(0.5)                       # the others are already knocking
        uid = wd.ship_unique_id(rank, world, lambda: bytes(range(128)), port=port, addr="127.0.0.1", timeout_s=60)
        assert uid == bytes(range(128)), uid
    """) % ROOT
    script = tmp_path / "ship.py"
    script.write_text(code)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port)]) for r in (1, 2, 0)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    assert wd.ship_unique_id(0, 1, lambda: b"x" * 128) == b"x" * 128


def test_comm_id_server_turns_strangers_away_and_serves_every_rank_until_acknowledged(tmp_path):
    """a port probe and a client of ANOTHER job must not take a real rank's place; a rank that never comes makes rank 0 give
    up after its timeout instead of hanging.  ADVICE r05: a rank counts as served when its acknowledgement has arrived -- a client that
    was cut off after the hello (here: one that hangs up without reading) does not count, the same rank asking again is served again,
    and a rank number outside the world is turned away."""
    import threading
    import time
    port = _free_port()
    got, errs = {}, {}

    def run(rank, job, key, delay=0.0, world=3, timeout_s=20):
        time.sleep(delay)
        try:
            got[key] = wd.ship_unique_id(rank, world, lambda: bytes(range(128)), port=port, addr="127.0.0.1", timeout_s=timeout_s, job=job)
        except Exception as e:   # noqa: BLE001
            errs[key] = e

    def probe():
        time.sleep(0.3)
        with socket.create_connection(("127.0.0.1", port), timeout=5) as c:   # says nothing sensible and hangs up
            c.sendall(b"GET / HTTP/1.0\r\n\r\n")

    def cut_off():   # rank 2 of the right job says hello and hangs up before it has read the id: it must NOT count as served
        import struct
        time.sleep(0.5)
        with socket.create_connection(("127.0.0.1", port), timeout=5) as c:
            c.sendall(wd._ID_MAGIC + struct.pack("<ii", 2, 3) + b"jobA".ljust(32, b"\0"))

    th = [threading.Thread(target=run, args=(0, "jobA", "r0")), threading.Thread(target=probe), threading.Thread(target=cut_off),
          threading.Thread(target=run, args=(1, "jobB", "stranger", 0.4)),          # same port, another job
          threading.Thread(target=run, args=(7, "jobA", "rank 7 of 3", 0.5)),       # right job, impossible rank
          threading.Thread(target=run, args=(1, "jobA", "r1", 0.6)), threading.Thread(target=run, args=(1, "jobA", "r1 again", 0.9)),
          threading.Thread(target=run, args=(2, "jobA", "r2", 1.6))]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert got.get("r0") == got.get("r1") == got.get("r2") == bytes(range(128)), (got.keys(), errs)
    assert isinstance(errs.get("stranger"), RuntimeError) and isinstance(errs.get("rank 7 of 3"), RuntimeError), errs
    assert got.get("r1 again") == bytes(range(128)) and set(got) == {"r0", "r1", "r1 again", "r2"}, (got.keys(), errs)
    # a rank that never shows up: rank 0 fails after its timeout
    port = _free_port()
    t0 = time.time()
    try:
        wd.ship_unique_id(0, 2, lambda: b"y" * 128, port=port, addr="127.0.0.1", timeout_s=1.0, job="jobC")
        assert False, "rank 0 returned although rank 1 never asked"
    except TimeoutError:
        assert time.time() - t0 < 10


def test_single_process_helpers_are_identity():
    import torch
    t = torch.tensor([3.0, 1.0])
    assert wd.allreduce_min_(t) is None and t.tolist() == [3.0, 1.0]
    assert wd.max_over_ranks(1.5, torch.device("cpu")) == 1.5 and wd.sum_over_ranks(7, torch.device("cpu")) == 7
    assert wd.shard_problems(10, 0, 1) == list(range(10)) and wd.shard_problems(10, 3, 4) == [3, 7]
    assert wd.env_rank() == (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def test_synth_grid_matches_oracle_generator():
    import oracle_lib as O
    from welding_robot_amd import synth
    for n, seed, pr in [(16, 1, 0.3), (48, 2025, 0.10)]:
        free, cx, cy, cz, p, wall = synth.synth_grid(n, seed, pr)
        og = O.synth_grid(n, seed, pr)
        assert np.array_equal(free, og.free) and np.array_equal(cx, og.cx) and p == 1.0 and wall == 0
    ids = synth.synth_weld_points(free, 48, 8)
    assert len(set(ids.tolist())) == 8 and np.all(free[ids] == 1)


def test_pairs_are_dealt_longest_first_over_end_point_groups():
    """deal_pairs (the rule of the drop-in C++ pair loop): every pair exactly once, loads balanced, whole end-point groups
    per rank while there are enough of them, longest searches first within a rank."""
    import numpy as np
    rs = np.random.RandomState(3)
    P = 64
    pts = rs.randint(0, 256, (P, 3))
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    w = [1 + int(np.abs(pts[i] - pts[j]).sum()) for i, j in pairs]
    for world in (1, 2, 3, 8):
        shards, load = wd.deal_pairs(pairs, w, world)
        assert sorted(k for sh in shards for k in sh) == list(range(len(pairs)))
        assert load == [sum(w[k] for k in sh) for sh in shards]
        assert max(load) <= 1.05 * (sum(w) / world), (world, load)          # 63 groups over <= 8 ranks balance to a few percent
        for sh in shards:                                                       # whole groups: an end point lives on one rank
            ends = {pairs[k][1] for k in sh}
            assert all((pairs[k][1] in ends) == (k in sh) for k in range(len(pairs)))
            assert w[sh[0]] == max(w[k] for k in sh)                            # the longest search of the rank leads
    # few points: single pairs are dealt, loads differ by at most one search
    pairs5 = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    w5 = [3, 9, 4, 7, 5, 8, 2, 6, 1, 10]
    shards, load = wd.deal_pairs(pairs5, w5, 3)
    assert sorted(k for sh in shards for k in sh) == list(range(10)) and max(load) - min(load) <= max(w5)
    assert wd.deal_pairs(pairs5, w5, 3) == (shards, load)


def test_a_batch_is_ordered_longest_first_in_each_half_of_its_slots():
    """order_batch: the slot order of ONE batch -- a permutation of the batch; each half of the slots (one pipelined group of the library) starts with
    its longest search and descends; the two halves get alternate entries of the sorted list (equal loads to within one search)."""
    import numpy as np
    rs = np.random.RandomState(5)
    for n in (1, 2, 7, 224):
        idx = list(rs.permutation(5000)[:n])
        w = {k: int(rs.randint(1, 700)) for k in idx}
        o = wd.order_batch(idx, w)
        assert sorted(o) == sorted(idx)
        h0, h1 = o[:n // 2], o[n // 2:]                              # group 0 = slots [0, n / 2), group 1 = the rest (wa_acs_run)
        for h in (h0, h1):
            assert all(w[h[i]] >= w[h[i + 1]] for i in range(len(h) - 1))
        if n >= 2:
            assert max(w[k] for k in idx) in (w[h0[0]], w[h1[0]])
            assert abs(sum(w[k] for k in h0) - sum(w[k] for k in h1)) <= max(w.values())
        assert wd.order_batch(idx, w) == o                            # deterministic
