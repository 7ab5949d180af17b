#!/usr/bin/env python3
"""Extract the hand-scheduled walk loop as assembled (hipcc -S, device only) and count its instructions.

    python tools/walk_isa.py [out.txt]        (default profiles/r06/walk_loop_isa.txt; no GPU needed)

The first line of the output is machine-readable -- bench.py derives `walk_step.instructions_per_step` from it
instead of carrying a number in its source:   # instructions_per_step: N  (4 steps per trip, M instructions per trip)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import build  # noqa: E402

KERNEL = "_Z10k_walk_devILb1ELb0ELb1ELb0ELb0ELb0EEv8WaAcsDev5WaRuniii"   # k_walk_dev<alpha 1, dense, touch loads, no rejoin watch, look-ahead>


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06", "walk_loop_isa.txt")
    asm = "/tmp/weldacs_walk_%d.s" % os.getpid()
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.hipcc()] + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, "weldacs.hip"), "-o", asm],
                          stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    os.unlink(asm)
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    top = next(i for i in range(start, len(lines)) if re.match(r"Lwa_top\d*:", lines[i]))
    end = next(i for i in range(top, len(lines)) if re.match(r"\s*s_branch\s+Lwa_top", lines[i]))
    body = lines[top:end + 1]
    ins = [l.strip() for l in body if l.strip() and not l.strip().endswith(":") and not l.strip().startswith((";", ".", "//"))]
    trip = len(ins) - 1                      # (the back edge belongs to the trip, not to a step)
    step_a = []
    for l in body[1:]:
        if re.match(r"Lwa_redo_b\d*:", l.strip()):
            break
        step_a.append(l.strip())
    # step 'a' runs from the top to the instruction before step b's head (v_add of the touch address + s_waitcnt precede Lwa_redo_b)
    n_a = len([l for l in step_a if l and not l.endswith(":")]) - 2
    n_vmem = len([l for l in ins if l.startswith(("global_load", "global_store", "buffer_load"))]) // 4
    n_br = len([l for l in ins if l.startswith("s_cbranch")]) // 4
    with open(out, "w") as f:
        f.write("# instructions_per_step: %d  (4 steps per trip, %d instructions per trip + 1 back edge; step 'a' alone: %d)\n" % (trip // 4, trip, n_a))
        f.write("# vector_memory_per_step: %d  conditional_branches_per_step: %d   (priced apart from the plain issue slots: profiles/r06/walk_trims.txt)\n" % (n_vmem, n_br))
        f.write("# the hand-scheduled general step of the ant walk (ACSRank_3D.hpp:134-193) as assembled for gfx950: %s, one trip = steps a-d\n" % KERNEL)
        f.write("\n".join(l.rstrip() for l in body) + "\n")
    print(open(out).readline().strip())


if __name__ == "__main__":
    main()
