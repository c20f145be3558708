#!/usr/bin/env python3
"""How well the batches in flight fill the chip, from a rocprofv3 --kernel-trace CSV of bench.py: over the middle of the run
(between the 25 % and 75 % quantile of the k_prepare launches) -- the share of time during which at least one chip-filling
kernel (k_prepare, k_small_accumulate, k_points_tables, k_static_accumulate) runs, their mean concurrency, the time no zk
kernel at all runs, and per kernel: launches, mean duration, queue ids.
usage: overlap.py <kernel_trace.csv>"""
import collections, csv, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
prep = [r for r in rows if r[2] == "k_prepare"]
lo, hi = prep[len(prep) // 4][0], prep[3 * len(prep) // 4][0]
win = [r for r in rows if r[1] > lo and r[0] < hi]
FILL = {"k_prepare", "k_small_accumulate", "k_points_tables", "k_static_accumulate"}


def coverage(sel):
    ev = []
    for s, e, _, _ in sel:
        ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
    ev.sort()
    cov = area = 0
    depth, last = 0, lo
    for t, d in ev:
        if depth > 0:
            cov += t - last
        area += depth * (t - last)
        depth += d; last = t
    return cov, area


span = hi - lo
cf, af = coverage([r for r in win if r[2] in FILL])
ca, aa = coverage([r for r in win if r[2].startswith("k_")])
n_prep = sum(1 for r in prep if lo <= r[0] < hi)
print("window %.3f ms, %d device batches started (%.3f ms per device batch)" % (span / 1e6, n_prep, span / 1e6 / max(n_prep, 1)))
print("chip-filling kernel active %.1f %% of the window, mean concurrency %.2f while active" % (100.0 * cf / span, af / max(cf, 1)))
print("any zk kernel active %.1f %% of the window, mean concurrency %.2f" % (100.0 * ca / span, aa / max(ca, 1)))
per = collections.defaultdict(list)
for s, e, n, q in win:
    per[n].append((e - s, q))
for n in sorted(per, key=lambda n: -sum(d for d, _ in per[n])):
    ds = [d for d, _ in per[n]]
    print("  %-24s %5d launches  mean %8.1f us  sum %9.3f ms  queues %s" % (n, len(ds), sum(ds) / len(ds) / 1e3, sum(ds) / 1e6, sorted({q for _, q in per[n]})))
