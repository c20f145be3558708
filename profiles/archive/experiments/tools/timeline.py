#!/usr/bin/env python3
"""Summarise overlap in a rocprofv3 kernel-trace CSV: per-queue lanes of the last few verify calls.
usage: timeline.py <kernel_trace.csv> [t_window_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
tr = [r for r in rows if r["Kernel_Name"].startswith("zk::k_transcript")]
# take a window in the middle of the timed loop
mid = tr[int(sys.argv[3]) if len(sys.argv) > 3 else len(tr) // 4]["s"]
sel = [r for r in rows if mid <= r["s"] < mid + win * 1e6]
qs = sorted({r["Queue_Id"] for r in sel})
print("queues in window:", qs, " all queues:", sorted({r["Queue_Id"] for r in rows}))
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("zk::", "")
    print("%9.3f %9.3f  q%-3s %s" % ((r["s"] - mid) / 1e6, (r["e"] - r["s"]) / 1e6, r["Queue_Id"], name))
