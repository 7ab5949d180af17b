#!/usr/bin/env python3
"""Cut a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --no-extras ...` at the TIMED REGION and hold it against the
JSON line of the same run: with --no-extras the command enqueues W warm-up generations and then K timed ones and
nothing else on the solver, so the timed region is dispatches [W, W + K) of each of the loop's three kernels.  Prints, per kernel, the
launches / total / average inside the window, the window's span from the first walk's start to the last apply's end, the sum of the
three kernels against ms_per_step x K, and the roofline figure the window's sweep launches give.

  python tools/timed_window_stats.py <rocprof output dir or kernel_trace.csv> <bench json line file> [--csv out.csv]
"""
import csv
import glob
import json
import os
import sys

LOOP = [("walk", "k_walk_dev"), ("sweep + rank + mark", "k_evap_rank_mark"), ("apply + replay table", "k_apply_table")]


def main():
    path, jpath = sys.argv[1], sys.argv[2]
    d = json.loads([l for l in open(jpath).read().splitlines() if l.startswith("{")][-1])
    K, W = int(d["steps"]), int(d["warmup"])
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    out = []
    series = {}
    t_first, t_last, total = None, None, 0.0
    for label, key in LOOP:
        mine = [(a, b) for a, b, k in rows if key in k]
        assert len(mine) >= W + K, "%s: %d dispatches in the trace, the command enqueued at least %d" % (key, len(mine), W + K)
        extra = len(mine) - (W + K)
        win = mine[W:W + K]
        us = [(b - a) / 1e3 for a, b in win]
        series[key] = us
        out.append(dict(kernel=key, role=label, launches_in_window=len(win), total_us=sum(us), avg_us=sum(us) / len(us), min_us=min(us), max_us=max(us),
                        dispatches_outside_window=W + extra))
        total += sum(us)
        t_first = win[0][0] if t_first is None else min(t_first, win[0][0])
        t_last = win[-1][1] if t_last is None else max(t_last, win[-1][1])
    span_us = (t_last - t_first) / 1e3
    ms_step = float(d["ms_per_step"])
    print("timed region: generations %d..%d (dispatches [%d, %d) of each loop kernel; %d warm-up generations in front)" % (0, K - 1, W, W + K, W))
    for r in out:
        print("  %-22s %-18s %3d launches  total %9.1f us  avg %7.2f us  (min %.2f, max %.2f); %d dispatches outside the window"
              % (r["role"], r["kernel"], r["launches_in_window"], r["total_us"], r["avg_us"], r["min_us"], r["max_us"], r["dispatches_outside_window"]))
    print("  sum of the three kernels in the window  %9.1f us = %.2f us per generation" % (total, total / K))
    print("  window span (first walk start -> last apply end) %9.1f us = %.2f us per generation" % (span_us, span_us / K))
    print("  bench.py: ms_per_step %.4f ms x %d = %.1f us   (kernels / wall %.3f, span / wall %.3f)" % (ms_step, K, ms_step * K * 1e3, total / (ms_step * K * 1e3), span_us / (ms_step * K * 1e3)))
    ok = total <= ms_step * K * 1e3 * 1.001
    print("  sum of kernels <= ms_per_step x K: %s" % ("yes" if ok else "NO"))
    sweep = out[1]
    alg = float(d["roofline"]["algorithmic_bytes_per_launch"])
    print("  roofline from the window alone: %.0f B / %.2f us = %.2f TB/s = %.3f of 8 TB/s   (bench.py's event-based frac: %.3f)"
          % (alg, sweep["avg_us"], alg / sweep["avg_us"] / 1e6, alg / sweep["avg_us"] / 1e6 / 8.0, d["roofline"]["frac"]))
    if "--ranges" in sys.argv:   # where the time goes, by generation of the search (the timed region starts at generation 0)
        print("  by generation range (average us per launch: walk / sweep + rank + mark / apply + table; longest walk launch of the range):")
        for lo, hi in ((0, 10), (10, 20), (20, 30), (30, 40), (40, 60), (60, 80), (80, 100), (100, 200), (200, 500)):
            if lo >= K:
                break
            hi = min(hi, K)
            w, f, a = (series[k][lo:hi] for _, k in LOOP)
            print("    generations %3d-%3d: %7.2f / %6.2f / %6.2f   sum %7.2f   longest walk %7.2f" % (lo, hi - 1, sum(w) / len(w), sum(f) / len(f), sum(a) / len(a),
                                                                                                    (sum(w) + sum(f) + sum(a)) / len(w), max(w)))
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
            w = csv.DictWriter(f, fieldnames=list(out[0]))
            w.writeheader()
            w.writerows(out)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
