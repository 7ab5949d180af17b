#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace CSV: per kernel name launches / total / average duration, the span from the first
start to the last end, the time during which at least one / at least two kernels were running, and per queue the busy time.

  python tools/trace_overlap.py <dir or kernel_trace.csv> [--skip-first N] [--from-kernel NAME]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
    rows.sort()
    rows = rows[skip:]
    if "--series" in sys.argv:   # per queue and kernel name: the durations in launch order (us), ten per line
        ser = defaultdict(list)
        for a, b, k, q in rows:
            ser[(q, k)].append((b - a) / 1e3)
        for (q, k), v in sorted(ser.items()):
            if len(v) < 8:
                continue
            print("q%s %s: %d launches, total %.1f us" % (q, k, len(v), sum(v)))
            for i in range(0, len(v), 10):
                print("   " + " ".join("%6.1f" % x for x in v[i:i + 10]))
        return
    if "--timeline" in sys.argv:   # the first N kernels behind the skipped ones: start / end in us relative to the first, queue, name
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        for a, b, k, q in rows[:n]:
            print("%10.1f %10.1f  %7.1f us  q%-3s %s" % ((a - rows[0][0]) / 1e3, (b - rows[0][0]) / 1e3, (b - a) / 1e3, q, k))
        return
    if not rows:
        print("no kernels")
        return
    by = defaultdict(lambda: [0, 0])
    byq = defaultdict(lambda: [0, 0])
    for a, b, k, q in rows:
        by[k][0] += 1
        by[k][1] += b - a
        byq[q][0] += 1
        byq[q][1] += b - a
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for a, b, _, _ in rows:
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    depth, last, busy1, busy2, busy3 = 0, ev[0][0], 0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth >= 3:
            busy3 += t - last
        depth += d
        last = t
    span = t1 - t0
    print("kernels %d  span %.3f ms  sum of durations %.3f ms  >=1 running %.3f ms (%.0f%%)  >=2 running %.3f ms (%.0f%%)  >=3 running %.3f ms (%.0f%%)" % (
        len(rows), span / 1e6, sum(v[1] for v in by.values()) / 1e6, busy1 / 1e6, 100 * busy1 / span, busy2 / 1e6, 100 * busy2 / span, busy3 / 1e6, 100 * busy3 / span))
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %6d launches  %9.3f ms  avg %8.2f us" % (k, n, t / 1e6, t / n / 1e3))
    for q, (n, t) in sorted(byq.items()):
        print("  queue %-6s %6d launches  busy %9.3f ms (%.0f%% of the span)" % (q, n, t / 1e6, 100 * t / span))


if __name__ == "__main__":
    main()
