#!/usr/bin/env python3
"""Timeline of the LAST zkgpu_tx_verify_batch call in a rocprofv3 --kernel-trace CSV of tools/tx_bench.py: per queue, the
kernels with start (ms, relative to the call's first kernel), duration and name.  usage: tx_trace.py <kernel_trace.csv> [window_ms]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
# the last call: walk back from the end while gaps stay below 2 ms
i = len(rows) - 1
while i > 0 and rows[i][0] - rows[i - 1][1] < 2_000_000 and rows[-1][1] - rows[i - 1][0] < win * 1e6:
    i -= 1
t0 = rows[i][0]
print("last burst: %d kernels, %.3f ms" % (len(rows) - i, (rows[-1][1] - t0) / 1e6))
for s, e, n, q in rows[i:]:
    print("%8.3f %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
