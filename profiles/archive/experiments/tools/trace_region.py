#!/usr/bin/env python3
"""The timed region of `bench.py --lean --steps 20` in a rocprofv3 --kernel-trace CSV: the two device batches that run side by
side (the pair of big k_transcript launches on different queues that start within 2 ms of each other), every kernel from the
first merge to the last bitmap: start, duration, end (us), queue, name.   usage: trace_region.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
tr = [i for i, r in enumerate(rows) if nm(r) == "k_transcript" and int(r["Grid_Size_X"]) >= 8192]
pair = None
for a, b in zip(tr, tr[1:]):
    if rows[a]["Queue_Id"] != rows[b]["Queue_Id"] and int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) < 2_000_000:
        pair = (a, b)
if not pair:
    sys.exit("no pair of concurrent device batches found")
i0 = pair[0]
while i0 > 0 and nm(rows[i0]) != "k_merge_inputs":
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
last = t0
for r in rows[i0:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if s > 12000 or (s - (last - t0) / 1e3 > 1500 and s > 3000):
        break
    last = int(r["End_Timestamp"])
    print("%8.1f %8.1f %8.1f  q%-3s %-26s grid %s" % (s, d, s + d, r["Queue_Id"], nm(r)[:26], r["Grid_Size_X"]))
print("region: %.1f us" % ((last - t0) / 1e3))
