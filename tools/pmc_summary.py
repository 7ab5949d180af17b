#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel averages of a kernel trace, or of one PMC counter.

    python tools/pmc_summary.py <dir> [counter]

Counter values FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half the bytes of
a wide (16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section): the `bytes_corrected`
column doubles FETCH_SIZE and takes WRITE_SIZE as is."""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    counter = sys.argv[2] if len(sys.argv) > 2 else None
    if counter:
        rows = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter:
                    rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        print("kernel,dispatches,avg_%s_KB,bytes_corrected_per_dispatch" % counter)
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            corr = avg * 1024 * (2 if counter == "FETCH_SIZE" else 1)
            print("%s,%d,%.3f,%.0f" % (k, len(v), avg, corr))
    else:
        rows = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("kernel,dispatches,avg_ns,total_ms")
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            print("%s,%d,%.1f,%.3f" % (k, len(v), sum(v) / len(v), sum(v) / 1e6))


if __name__ == "__main__":
    main()
