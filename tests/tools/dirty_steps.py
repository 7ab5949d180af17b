#!/usr/bin/env python3
"""How many ant steps stand on a voxel that has EVER received a deposit?  (CPU only; round 5.)  A "clean" voxel's six edges all hold the
same value p0 * rho^t, so a walk could decide its step there from admissibility bits and one scalar, without fetching the voxel's
24-byte record -- IF most steps stood on clean voxels.  Counted with an instrumented COPY of the oracle (oracle/weld_oracle.c is copied
to a temp directory and patched there: a byte per voxel set by update_pheromone's loop, a counter in the walk; the oracle itself is not
touched), DEV mode, BASELINE-shaped searches.

    python tests/tools/dirty_steps.py [out.txt]"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(tmp):
    s = open(os.path.join(ROOT, "oracle", "weld_oracle.c")).read()
    n = 0

    def rep(old, new):
        nonlocal s, n
        assert old in s, old
        s = s.replace(old, new, 1)
        n += 1
    rep('#include <time.h>\n', '#include <time.h>\n#include <stdio.h>\nuint8_t *g_dirty; long long g_cnt[2];\n'
        'void wo_dirty_init(long long n){ free(g_dirty); g_dirty = calloc(n,1); g_cnt[0]=g_cnt[1]=0; }\n')
    rep('        float info[26];\n', '        if (g_dirty) g_cnt[g_dirty[cur]]++;\n        float info[26];\n')
    rep("                int k = a->choice[i + 1];\n", "                int k = a->choice[i + 1];\n                if (g_dirty) g_dirty[v] = 1;\n")
    rep("        if (trace_bestL) trace_bestL[g] = s->best.L;",
        '        if (g_dirty) { printf("gen %d clean %lld dirty %lld best %g\\n", g, g_cnt[0], g_cnt[1], s->best.L); fflush(stdout); g_cnt[0]=g_cnt[1]=0; }\n'
        "        if (trace_bestL) trace_bestL[g] = s->best.L;")
    open(os.path.join(tmp, "weld_oracle.c"), "w").write(s)
    subprocess.check_call(["cp", os.path.join(ROOT, "oracle", "weld_oracle.h"), tmp])
    lib = os.path.join(tmp, "libdirty.so")
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-o", lib, os.path.join(tmp, "weld_oracle.c"), "-lm"])
    return lib


CHILD = r"""
import sys, ctypes as C
import numpy as np
import oracle_lib as O
L = O.lib()
for n, ants, gens, predict, label in ((64, 24, 150, 24 / 0.35, "64^3 pair search, 24 ants (adaptive colony), 150 generations"),
                                      (128, 256, 60, 731.43, "128^3 corner to corner, 256 ants fixed (C3), 60 generations")):
    og = O.synth_grid(n, seed=2024, occ_prob=0.10)
    free = np.nonzero(og.free)[0]
    rs = np.random.RandomState(1)
    sid, eid = (int(rs.choice(free)), int(rs.choice(free))) if ants == 24 else (int(free[0]), int(free[-1]))
    L.wo_dirty_init(C.c_longlong(n ** 3))
    print("== " + label, flush=True)
    O.Acs(og).solve(sid, eid, gens, predict, fixed_colony=ants if ants == 256 else 0, mode=O.DEV, seed=7, stream=5)
"""


def main():
    with tempfile.TemporaryDirectory() as tmp:
        lib = build(tmp)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, cwd=os.path.join(ROOT, "tests"), env=dict(os.environ, WELD_ORACLE_LIB=lib))
        assert r.returncode == 0, r.stderr[-2000:]
    acc, cur = {}, None
    for l in r.stdout.splitlines():
        if l.startswith("=="):
            cur = l[3:]
            acc[cur] = []
            continue
        m = re.match(r"gen (\d+) clean (\d+) dirty (\d+) best (\S+)", l)
        if m:
            acc[cur].append(tuple(float(x) for x in m.groups()))
    out = ["# steps on voxels that have ever received a deposit (\"dirty\") -- tests/tools/dirty_steps.py, an instrumented copy of the oracle, DEV mode"]
    for k, v in acc.items():
        out.append(k)
        for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 150)):
            sel = [x for x in v if lo <= x[0] < hi]
            if sel:
                c, d = sum(x[1] for x in sel), sum(x[2] for x in sel)
                out.append("  generations %3d-%3d: %8.0f steps per generation, %5.1f %% of them on a dirty voxel, best %g" % (lo, hi - 1, (c + d) / len(sel), 100.0 * d / (c + d), sel[-1][3]))
        c, d = sum(x[1] for x in v), sum(x[2] for x in v)
        out.append("  all %d steps: %.1f %% on a dirty voxel" % (c + d, 100.0 * d / (c + d)))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text, end="")


if __name__ == "__main__":
    main()
