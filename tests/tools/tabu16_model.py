#!/usr/bin/env python3
"""Can a 16-bit tabu entry -- 12-bit quotient of a bijective hash + 4-bit probe displacement (csrc/acs_walk.hpp WaTabu) -- name every voxel an ant
has visited?  Only while no insert lands more than 13 slots from its home slot.  For uniformly random keys linear probing needs displacements of
20-50 at the loads a walk reaches; a lattice walk's ids are near-sequential, and a golden-ratio multiplier of the ids' own bit width spreads them
almost perfectly.  Measured on the ORACLE's own ant paths (first two generations of far-apart pair searches, 24 ants, DEV mode): the displacement
of every insert into a 2^12-slot table, for (a) the 32-bit multiplier's low B bits and (b) K_B = odd(2^B / golden ratio), B = bits of the grid's ids.
CPU only (under tests/: it runs the oracle).      python tests/tools/tabu16_model.py [out.txt]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

PHI = 0.6180339887498949


def stats(paths, log2, B, KB):
    size = 1 << log2
    mask = size - 1
    mx = over = tot = walks_over = 0
    dsum = 0
    for p in paths:
        tab = np.full(size, -1, np.int64)
        p = p[:int(0.75 * size)]
        wo = False
        for key in p:
            h = ((int(key) * KB) & ((1 << B) - 1)) >> (B - log2)
            d = 0
            while tab[(h + d) & mask] != -1:
                d += 1
            tab[(h + d) & mask] = int(key)
            mx = max(mx, d)
            tot += 1
            dsum += d
            if d >= 14:
                over += 1
                wo = True
        walks_over += wo
    return "largest displacement %2d, mean %.3f, inserts >= 14 slots from home %d of %d, walks with one %d of %d" % (mx, dsum / tot, over, tot, walks_over, len(paths))


def main():
    out = ["# tests/tools/tabu16_model.py: displacement of every tabu insert along the oracle's own ant paths, table of 2^12 slots (3 072 = the spill threshold at most)"]
    for n, B in ((256, 24), (128, 21)):
        og = O.synth_grid(n, seed=2024, occ_prob=0.10)
        free = np.nonzero(og.free)[0]
        rs = np.random.RandomState(5)
        allp = []
        for trial in range(4):
            while True:
                sid, eid = int(rs.choice(free)), int(rs.choice(free))
                d = abs(sid % n - eid % n) + abs((sid // n) % n - (eid // n) % n) + abs(sid // (n * n) - eid // (n * n))
                if d > 1.6 * n:
                    break
            a = O.Acs(og)
            a.solve(sid, eid, 2, 24 / 0.35, fixed_colony=0, mode=O.DEV, seed=7, stream=trial)
            allp += [np.asarray(p) & 0x1fffffff for p in a.last_paths()]
        lens = [len(p) for p in allp]
        out.append("%d^3 (%d-bit ids): %d walks, %d nodes at most, %.0f on average" % (n, B, len(allp), max(lens), sum(lens) / len(lens)))
        out.append("   2 654 435 761 (2^32 / golden ratio), low %d bits : %s" % (B, stats(allp, 12, B, 2654435761 & ((1 << B) - 1))))
        out.append("   odd(2^%d / golden ratio)                        : %s" % (B, stats(allp, 12, B, int((1 << B) * PHI) | 1)))
    out.append("# (uniformly random 24-bit keys, 3 072 inserts into 2^12 slots: largest displacement ~50, mean 4.4 in the last quarter)")
    out.append("# On the device (profiles/r06/tab16.txt): 36 of 432 000 walks of a 256^3 batch meet an insert that would land 14 slots from home and spill to their bitmap;")
    out.append("# a LOOKUP is decided by its first fourteen slots whatever it finds there.")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text, end="")


if __name__ == "__main__":
    main()
