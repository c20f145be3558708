#!/usr/bin/env python3
"""Turn the raw output of tools/profile_bench.sh <tag> (under gpurun_out/) into the committed summaries:
  profiles/<tag>_bench_kernel_stats.csv            rocprofv3 --stats table of the default bench
  profiles/<tag>_bench_kernel_trace_summary.txt    per-kernel durations of the last dispatches (= timed steps)
  profiles/<tag>_pmc_<PASS>.txt                    per kernel and counter: dispatches, last value, mean
  profiles/pmc_traffic.json                        HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B)
usage: python tools/collect_profiles.py <tag>"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles")
go = os.path.join(ROOT, "gpurun_out")


def short(name):
    return name.split("(")[0].replace("zk::", "")


stats = glob.glob(os.path.join(go, "prof_%s" % tag, "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out, "%s_bench_kernel_stats.csv" % tag))
trace = glob.glob(os.path.join(go, "prof_%s" % tag, "*", "*_kernel_trace.csv"))
if trace:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), trace[0], "--last", "30"],
                         capture_output=True, text=True).stdout
    open(os.path.join(out, "%s_bench_kernel_trace_summary.txt" % tag), "w").write(txt.replace(go + "/", "gpurun_out/"))
fetch, write = {}, {}
for pas in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
    files = glob.glob(os.path.join(go, "pmc_%s_%s" % (tag, pas), "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        agg.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    with open(os.path.join(out, "%s_pmc_%s.txt" % (tag, pas)), "w") as f:
        f.write("# rocprofv3 --pmc (tools/profile_bench.sh %s), bench.py --inflight 1 --steps 4; per kernel: counter, dispatches, "
                "last value, mean\n" % tag)
        for k, cs in agg.items():
            for c, v in sorted(cs.items()):
                f.write("%-30s %-22s %4d %16.1f %16.1f\n" % (k[:30], c, len(v), v[-1], sum(v) / len(v)))
                # steady-state mean: the dispatches of the 4 timed steps and the solo pass (skip one-time launches)
                tail = v[-8:] if len(v) >= 8 else v
                if c == "FETCH_SIZE":
                    fetch[k] = sum(tail) / len(tail)
                if c == "WRITE_SIZE":
                    write[k] = sum(tail) / len(tail)
traffic = {"_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), the gfx950 correction of "
                    "MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B fabric requests as 64 B.  Separate --pmc passes "
                    "(tools/profile_bench.sh %s, bench.py --inflight 1), mean over the last <= 8 dispatches of each kernel.  "
                    "k_static_accumulate runs twice per batch when transactions are checked in groups (the group launch and "
                    "the re-check of failed groups): the figure is the mean of the two.  Its access shape (one 96-B row "
                    "per lane and addition out of a 26 GB table) is not one the guide calibrated: treat as +-2x." % tag}
for k in sorted(set(fetch) | set(write)):
    traffic[k] = int(1024 * (2 * fetch.get(k, 0.0) + write.get(k, 0.0)))
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("wrote", sorted(os.listdir(out)))
