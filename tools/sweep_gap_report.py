#!/usr/bin/env python3
"""Per-mode summary of tools/ubench/sweep_gap_pmc under rocprofv3 (round 6, VERDICT r05 task 4).

    python tools/sweep_gap_report.py <dir> <mode> [counter ...]

<dir> holds one rocprofv3 output directory per pass: <dir>/<mode>_trace (--kernel-trace) and <dir>/<mode>_<counter> (--pmc <counter>).
Sweep dispatches are taken in dispatch order; the first 8 (warm-up, back to back in every mode) are dropped.  Prints the average
duration of the sweeps, the gap between a sweep's start and the end of the dispatch in front of it, and per-dispatch counter averages
(GRBM_GUI_ACTIVE also as effective clock = value / 8 XCDs / duration, the guide's formula)."""
import csv
import glob
import os
import sys


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    base, mode = sys.argv[1], sys.argv[2]
    counters = sys.argv[3:]
    tr = rows(os.path.join(base, mode + "_trace"), "*kernel_trace.csv")
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    sw = [i for i, r in enumerate(tr) if r["Kernel_Name"].startswith("k_sweep")]
    dur = {0: [], 1: []}
    gap = {0: [], 1: []}
    for n, i in enumerate(sw):
        if n < 8:
            continue
        r = tr[i]
        behind_sweep = i > 0 and tr[i - 1]["Kernel_Name"].startswith("k_sweep")
        k = 1 if (mode == "idle_b2b2" and behind_sweep) else 0
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if i > 0:
            gap[k].append(int(r["Start_Timestamp"]) - int(tr[i - 1]["End_Timestamp"]))
    for k in (0, 1):
        if dur[k]:
            d = sorted(dur[k])
            print("%-10s kind %d  sweeps %3d  duration avg %7.0f ns  median %7.0f  min %7.0f   start - previous dispatch's end: avg %6.0f ns (min %d)"
                  % (mode, k, len(d), sum(d) / len(d), d[len(d) // 2], d[0], sum(gap[k]) / max(1, len(gap[k])), min(gap[k]) if gap[k] else 0))
    avg_dur = sum(dur[0]) / max(1, len(dur[0]))
    for c in counters:
        cr = rows(os.path.join(base, mode + "_" + c), "*counter_collection.csv")
        names = sorted(set(r["Counter_Name"] for r in cr))
        for nm in names:
            v = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in cr if r["Counter_Name"] == nm and r["Kernel_Name"].startswith("k_sweep")]
            v.sort()
            v = [x for _, x in v[8:]]
            if not v:
                continue
            a = sum(v) / len(v)
            extra = ""
            if nm == "GRBM_GUI_ACTIVE":
                extra = "   -> %.3f GHz over the traced duration of %.0f ns (value / 8 / duration; reads high on short dispatches, compare across modes)" % (a / 8 / avg_dur, avg_dur)
            print("%-10s %-34s avg per sweep dispatch %14.1f  (%d dispatches)%s" % (mode, nm, a, len(v), extra))


if __name__ == "__main__":
    main()
