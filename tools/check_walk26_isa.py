#!/usr/bin/env python3
"""Build-time check of what the 26-neighbour fast loop (k_walk_dev26, csrc/acs_nb26.hpp) relies on but the compiler does not promise
(ADVICE r03): its next-step record loads are ISSUED by one inline statement and WAITED for by a later one (`s_waitcnt vmcnt(4)`), and its
touch loads land in v250..v253, which only the statement's clobber list names.  Checked on the assembled kernel:

  1. v250..v253 appear in no instruction of the kernel other than the touch loads themselves;
  2. from EVERY load that is not waited for on the spot -- the record-load pairs, and single loads such as the lazily evaporated field's stamp --,
     along EVERY control-flow path, no instruction reads or writes its destination registers before an `s_waitcnt vmcnt(N)` has been passed with
     N <= the vector-memory instructions issued behind it on that path (no copy, no phi move, no spill, no younger load in between);
  3. apart from the clobbered v250..v253 the kernel's own registers stay far below them.

    python tools/check_walk26_isa.py [file.s]        (no GPU needed; without a file the library's device code is assembled with hipcc -S)
Exit status 0 and a one-line summary, or 1 and the offending instructions."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from welding_robot_amd import build  # noqa: E402

KERNEL_RE = r"^(_Z\d*k_walk_dev26\w*):"   # (the mangled name follows the signature: found by pattern, not spelled out)
TOUCH = {250, 251, 252, 253}


def assemble():
    asm = "/tmp/weldacs_walk26_%d.s" % os.getpid()
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.hipcc()] + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, "weldacs.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    os.unlink(asm)
    return text


def vregs(operand_text):
    """VGPR indices an operand string names: v12, v[8:11]"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(text):
    """every instantiation of k_walk_dev26 (dense field, lazily evaporated field) is checked; the summary describes the first"""
    lines = text.split("\n")
    hits = [i for i, l in enumerate(lines) if re.match(KERNEL_RE, l)]
    if not hits:
        return ["no k_walk_dev26 kernel in the device code"], {}
    problems, info = [], None
    for start in hits:
        p, i = check_kernel(lines, start)
        problems += ["%s: %s" % (lines[start].split(":")[0][:40], x) for x in p]
        info = info or i
    info["kernels"] = len(hits)
    return problems, info


def check_kernel(lines, start):
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins, labels = [], {}
    for l in lines[start + 1:end]:
        t = l.split(";")[0].strip()
        if not t or t.startswith((".", "//")) and not t.endswith(":"):
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        ins.append(t)
    problems = []
    # 1. the touch registers
    touch_sites = 0
    for i, t in enumerate(ins):
        used = vregs(t.split(None, 1)[1] if " " in t else "")
        if used & TOUCH:
            if t.startswith("global_load_dword v25") and len(used & TOUCH) == 1:
                touch_sites += 1
            else:
                problems.append("v250..v253 used outside the touch loads: [%d] %s" % (i, t))
    if touch_sites == 0 or touch_sites % 4:
        problems.append("expected groups of four touch loads, found %d" % touch_sites)
    # 2. issue ... wait.  EVERY load of the kernel whose result is not waited for on the spot -- pairs, and singles such as the lazily evaporated
    # field's stamp load -- is followed along every control-flow path: vector memory returns in order, so its registers are
    # good once an s_waitcnt vmcnt(N) has been passed with N <= the vector-memory instructions issued behind it on that path; until then no
    # instruction may read or write them (no copy, no phi move, no spill, no younger load into the same register)
    def waits_vm(t):
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        return int(m.group(1)) if m else None
    def is_vmem(t):
        return t.startswith(("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store", "flat_atomic", "scratch_"))
    pairs = singles = 0
    single_sites = []
    for i in range(len(ins)):
        if not ins[i].startswith("global_load_dword") or vregs(ins[i].split(",")[0]) & TOUCH:
            continue
        regs = vregs(ins[i].split(",")[0])
        # (a load is followed on its own: the loads behind it in the same run are among its `younger` ones -- the compiler's partial waits on a long
        #  run, vmcnt(20) in front of the use of the run's first results, pass the same rule as the loop's exact vmcnt(4))
        deferred_here = False
        seen, todo = set(), [(i + 1, 0)]
        bad = None
        while todo and bad is None:
            k, younger = todo.pop()
            while k < len(ins) and (k, min(younger, 64)) not in seen:
                seen.add((k, min(younger, 64)))
                t = ins[k]
                n = waits_vm(t)
                if n is not None and n <= younger:
                    if n > 0:
                        deferred_here = True
                    break
                op = t.split(None, 1)
                if len(op) > 1 and vregs(op[1]) & regs:
                    bad = "records loaded at [%d] (v%s) touched before their wait: [%d] %s" % (i, sorted(regs), k, t)
                    break
                if is_vmem(t):
                    younger += 1
                if t.startswith("s_endpgm"):
                    break
                m = re.match(r"s_c?branch\w*\s+(\S+)", t)
                if m and m.group(1) in labels:
                    todo.append((labels[m.group(1)], younger))
                    if t.startswith("s_branch"):
                        break
                k += 1
        if bad:
            problems.append(bad)
        elif deferred_here:
            # (the loop's own: a deferred wait of exactly four -- the touch loads -- behind a record pair, or behind the pair + the lazy field's stamp)
            if i + 1 < len(ins) and ins[i + 1].startswith("global_load_dword") and not (vregs(ins[i + 1].split(",")[0]) & TOUCH):
                pairs += 1
            else:
                singles += 1
                single_sites.append(ins[i])
    if pairs == 0:
        problems.append("no record-load pair with a deferred wait found: has the loop changed?")
    # 3. register budget
    own = set()
    for t in ins:
        op = t.split(None, 1)
        if len(op) > 1:
            own |= vregs(op[1]) - TOUCH
    top = max(own) if own else -1
    if top >= 200:
        problems.append("the kernel's own VGPRs reach v%d: too close to the hard-coded touch registers" % top)
    return problems, dict(instructions=len(ins), deferred_pairs=pairs, deferred_singles=singles, single_sites=single_sites, touch_loads=touch_sites, highest_own_vgpr=top)


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else assemble()
    problems, info = check(text)
    if problems:
        print("k_walk_dev26 ISA check FAILED:")
        for p in problems:
            print("  " + p)
        return 1
    print("k_walk_dev26 ISA check ok (%(kernels)d instantiation(s)): %(instructions)d instructions, %(deferred_pairs)d record-load pairs with a deferred wait, "
          "%(deferred_singles)d single loads with one, %(touch_loads)d touch loads in v250..v253, highest VGPR of its own v%(highest_own_vgpr)d" % info)
    return 0


if __name__ == "__main__":
    sys.exit(main())
