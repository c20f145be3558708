#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_breakdown.sh <tag> -> profiles/<tag>_pmc_breakdown.txt"""
import collections, csv, glob, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(root, "gpurun_out", "pmcx_%s_*" % tag, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").split("<")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
out = ["# counters per launch (mean over the launches of `bench.py --solo --steps 6`, device batches of 10240 transactions), tools/pmc_breakdown.sh %s" % tag]
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0) / max(n[k].get("SQ_INSTS_VALU", 1), 1)):
    if not k.startswith("k_") or k in ("k_spin", "k_tbl_multiples", "k_tbl_base"):
        continue
    out.append(k)
    for c in sorted(tot[k]):
        out.append("    %-28s %14.4g" % (c, tot[k][c] / n[k][c]))
path = os.path.join(root, "profiles", "%s_pmc_breakdown.txt" % tag)
open(path, "w").write("\n".join(out) + "\n")
print("wrote", path)
