#!/usr/bin/env python3
"""Kernel timeline of the LAST burst of device batches in a rocprofv3 --kernel-trace CSV of `bench.py --lean` (the solo pass runs
last: pass --skip-solo to cut the serial launches at the end off by their signature -- k_small_accumulate launches that do not
overlap anything).  usage: bench_timeline.py <kernel_trace.csv> [window_ms=8]
Prints the kernels of the timed region's device batches: start (ms), duration, end, queue, name, grid."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Grid_Size_X", "?")))
rows.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
# `bench.py --lean --steps 20 --warmup 5` merges tickets into device batches: 5 (priming, one per lane) + 1 (the warm-up's five
# tickets) + 2 (THE TIMED STEPS) + 2 (the same steps again with events) -- one k_merge_inputs launch each; the solo pass has none
merges = [i for i, r in enumerate(rows) if r[2] == "k_merge_inputs"]
if len(merges) < 8:
    sys.exit("expected at least 8 k_merge_inputs launches, found %d" % len(merges))
first = merges[6]
t0 = rows[first][0]
t_end = rows[merges[8]][0] if len(merges) > 8 else t0 + int(win * 1e6)
sel = [r for r in rows[first:] if r[0] < t_end]
print("the timed steps: two device batches, %d kernels, %.3f ms from the first merge to the last kernel's end" % (len(sel), (max(x[1] for x in sel) - t0) / 1e6))
for s_, e, n, q, g in sel:
    print("%8.3f %8.3f %8.3f  q%-3s %-24s %s" % ((s_ - t0) / 1e6, (e - s_) / 1e6, (e - t0) / 1e6, q, n[:24], g))
