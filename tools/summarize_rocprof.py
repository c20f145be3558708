#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, calls / mean / median / min / max (us),
optionally restricted to the last N dispatches of each kernel (= the timed steps of bench.py).

  python tools/summarize_rocprof.py gpurun_out/prof/.../*_kernel_trace.csv [--last 20] > profiles/rNN_....txt
"""
import csv
import statistics
import sys


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            per.setdefault(name, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                             row.get("VGPR_Count", ""), row.get("LDS_Block_Size", ""),
                                             row.get("Grid_Size_X", ""), row.get("Workgroup_Size_X", "")))
    rows = []
    for name, xs in per.items():
        xs.sort()
        if last:
            xs = xs[-last:]
        d = [x[1] / 1e3 for x in xs]
        rows.append((sum(d), name, len(d), statistics.mean(d), statistics.median(d), min(d), max(d), xs[-1][2], xs[-1][3], xs[-1][4], xs[-1][5]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("# %s%s" % (path, " (last %d dispatches per kernel)" % last if last else ""))
    print("%-28s %6s %10s %10s %10s %10s %6s %5s %6s %9s %5s" % ("kernel", "calls", "mean_us", "median_us", "min_us", "max_us", "pct", "vgpr", "lds", "grid", "wg"))
    for r in rows:
        print("%-28s %6d %10.1f %10.1f %10.1f %10.1f %6.2f %5s %6s %9s %5s" % (r[1][:28], r[2], r[3], r[4], r[5], r[6], 100 * r[0] / tot, r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main()
