#!/usr/bin/env python3
"""Soak of the round-4 mechanisms against each other: random batches (1..8 searches per solver, 6 or 26 neighbours, dense or lazy) run
(a) plainly -- one stream, stragglers off, one wa_acs_run call -- and (b) with a random number of pipelined groups, the per-slot straggler
hand-over on, the run cut into random pieces with or without reads in between (chained calls hand over in their last generation and a
drain launch finishes those stragglers when a read comes first).  Everything observable must be equal: per-generation trace (best cost,
steps, finite ants), every ant of the last generation, the best path, the whole field.  Which ants are handed over depends on timing, so
repeated trials walk different code paths.       python tools/pipeline_soak.py [trials] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from welding_robot_amd import api  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def state(s, P, iters):
    out = []
    for q in range(P):
        t = s.trace(q)
        L, lens = s.ants(q)
        cost, path, _ = s.result(q)
        out.append((t["steps"][:iters].copy(), t["finite"][:iters].copy(), bits(t["bestL"][:iters]).copy(), bits(L).copy(), lens.copy(), bits(cost).copy(), path.copy(),
                    bits(s.pheromone(q)).copy()))
    return out


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    ctx = api.Context(0)
    mism, handed = 0, 0
    for trial in range(trials):
        nx, ny, nz = (int(rs.randint(20, 52)) for _ in range(3))
        n = nx * ny * nz
        free = (rs.uniform(size=n) >= float(rs.choice([0.0, 0.1, 0.2]))).astype(np.uint8)
        free[0] = free[-1] = 1
        ax = [np.arange(k, dtype=np.float32) for k in (nx, ny, nz)]
        g = api.Grid.from_occupancy(ctx, free, ax[0], ax[1], ax[2], 1.0, 0)
        kind = str(rs.choice(["dense", "dense", "nb26", "lazy"]))
        P, ants, iters = int(rs.randint(1, 9)), int(rs.choice([24, 64, 96, 200])), int(rs.randint(4, 24))
        fr = np.flatnonzero(free)
        starts = [int(rs.choice([0, int(rs.choice(fr))])) for _ in range(P)]
        ends = [int(rs.choice([n - 1, int(rs.choice(fr))])) for _ in range(P)]
        streams = [int(v) for v in rs.randint(0, 10000, P)]
        p = api.default_params(max_iteration=iters, predict=float(nx + ny + nz), fixed_colony=ants, rng_mode=api.RNG_DEV, seed=int(rs.randint(1, 1 << 30)))
        res = []
        for variant in (0, 1):
            s = api.AcsSolver(ctx, g, n_slots=P, max_colony=ants, neighbourhood=26 if kind == "nb26" else 6, lazy=kind == "lazy")
            s.init_pheromone(1.0)
            if variant == 0:
                s.set_pipeline(1)
                s.set_stragglers(0)
                s.solve(p, starts, ends, streams=streams)
            else:
                s.set_pipeline(int(rs.randint(0, P + 1)))
                s.begin(p, starts, ends, streams=streams)
                done = 0
                while done < iters:
                    c = int(min(iters - done, rs.randint(1, 7)))
                    s.run(c)
                    done += c
                    if rs.rand() < 0.4:
                        s.ants(int(rs.randint(0, P)))          # a read between calls: must be complete whenever it comes
                s.sync()
                handed += sum(s.straggler_counters(q)[0] for q in range(P))
                assert all(h == r for h, r in (s.straggler_counters(q) for q in range(P)))
            res.append(state(s, P, iters))
            s.close()
        for q in range(P):
            for a, b in zip(res[0][q], res[1][q]):
                if not np.array_equal(a, b):
                    mism += 1
                    print("MISMATCH trial %d kind %s P %d slot %d" % (trial, kind, P, q))
                    break
        g.close()
    print("trials %d, mismatches %d, ants handed over in total %d" % (trials, mism, handed))
    return 1 if mism else 0


if __name__ == "__main__":
    sys.exit(main())
