#!/usr/bin/env python3
"""Print a window of a ZKGPU_TIMELINE dump (ctx name start_ms end_ms), sorted by start.
usage: timeline2.py <file> [t0_ms] [len_ms]"""
import sys
rows = []
for line in open(sys.argv[1]):
    c, name, a, b = line.split()
    rows.append((float(a), float(b), c, name))
rows.sort()
ctxs = {c: i for i, c in enumerate(sorted({r[2] for r in rows}))}
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else rows[len(rows) // 2][0]
ln = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
for a, b, c, name in rows:
    if t0 <= a < t0 + ln:
        print("%9.3f %7.3f  c%d %s" % (a - t0, b - a, ctxs[c], name))
