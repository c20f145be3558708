#!/usr/bin/env python3
"""The last single 2^20 multiplication in a rocprofv3 --kernel-trace CSV of tools/msm_bench.py: every kernel from its first
k_decompress_pre to its k_window_partials -- start, duration, end (us), queue, name.   usage: msm_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
ends = [i for i, r in enumerate(rows) if nm(r) == "k_window_partials"]
if not ends:
    sys.exit("no k_window_partials in the trace")
i1 = ends[-1]
t1 = int(rows[i1]["End_Timestamp"])
# its kernels: everything that started within 4 ms before the end
sel = [r for r in rows[:i1 + 1] if t1 - int(r["Start_Timestamp"]) < 4_000_000]
first = [i for i, r in enumerate(sel) if nm(r) in ("k_decompress_pre", "k_part_hist")]
sel = sel[first[0] if not [i for i in first if i > 0 and int(sel[i]["Start_Timestamp"]) - int(sel[i - 1]["End_Timestamp"]) > 200_000] else
          [i for i in first if i > 0 and int(sel[i]["Start_Timestamp"]) - int(sel[i - 1]["End_Timestamp"]) > 200_000][-1]:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%8.1f %8.1f %8.1f  q%-3s %-26s grid %s" % (s, d, s + d, r["Queue_Id"], nm(r)[:26], r["Grid_Size_X"]))
print("region: %.1f us" % ((t1 - t0) / 1e3))
