#!/usr/bin/env python3
"""Turn the raw output of tools/profile_bench.sh <tag> (under gpurun_out/) into the committed summaries:
  profiles/<tag>_bench_kernel_stats.csv            rocprofv3 --stats table of the timed bench (batches in flight)
  profiles/<tag>_bench_kernel_trace_summary.txt    per-kernel durations of the last dispatches (= timed steps)
  profiles/<tag>_solo_kernel_stats.csv             rocprofv3 --stats table of `bench.py --solo` (each kernel alone on the chip)
  profiles/<tag>_msm_kernel_stats.csv              ... of the 2^20 MSM pipeline (tools/msm_bench.py)
  profiles/<tag>_prover_kernel_stats.csv           ... of the device prover (tools/prover_profile.py: 2048 cloak proofs per call)
  profiles/<tag>_{bench,solo,msm}.json             the JSON records those commands printed
  profiles/<tag>_pmc_<PASS>.txt, <tag>_pmcmsm_<PASS>.txt   per kernel and counter: dispatches, last value, mean
  profiles/pmc_traffic.json                        HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B)
  profiles/pmc_valu.json                           SQ_INSTS_VALU (wave instructions) per launch
usage: python tools/collect_profiles.py <tag>"""
import collections, csv, glob as _glob, json, os, shutil, subprocess, sys


class glob:     # gpurun MERGES what a call wrote into gpurun_out/: a directory may hold the files of earlier runs too -- newest only
    @staticmethod
    def glob(pattern):
        files = sorted(_glob.glob(pattern), key=os.path.getmtime)
        return files[-1:]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles")
go = os.path.join(ROOT, "gpurun_out")


def short(name):
    n = name.split("(")[0].replace("zk::", "")
    if n.startswith("void "):           # template instantiations are printed with their return type: void k_group_combine<false>
        n = n[5:]
    return n.split("<")[0]


for kind in ("prof", "solo", "msm", "prover", "proverprog", "tx"):
    label = {"prof": "bench", "solo": "solo", "msm": "msm", "prover": "prover", "proverprog": "proverprog", "tx": "tx"}[kind]
    txt_out = os.path.join(go, "%s_%s.txt" % (kind, tag))
    if os.path.exists(txt_out) and os.path.getsize(txt_out):
        shutil.copy(txt_out, os.path.join(out, "%s_%s_output.txt" % (tag, label)))
    stats = glob.glob(os.path.join(go, "%s_%s" % (kind, tag), "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, label)))
    trace = glob.glob(os.path.join(go, "%s_%s" % (kind, tag), "*", "*_kernel_trace.csv"))
    if trace:
        txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), trace[0], "--last", "30"],
                             capture_output=True, text=True).stdout
        open(os.path.join(out, "%s_%s_kernel_trace_summary.txt" % (tag, label)), "w").write(txt.replace(go + "/", "gpurun_out/"))
    for src in ("%s_%s_bench.json" % (kind, tag), "%s_%s.json" % (kind, tag)):
        p = os.path.join(go, src)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(out, "%s_%s.json" % (tag, label)))
fetch, write, valu = {}, {}, {}
for prefix in ("pmc", "pmcmsm"):
    for pas in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
        files = glob.glob(os.path.join(go, "%s_%s_%s" % (prefix, tag, pas), "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(files[0])):
            agg.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        with open(os.path.join(out, "%s_%s_%s.txt" % (tag, prefix, pas)), "w") as f:
            f.write("# rocprofv3 --pmc (tools/profile_bench.sh %s), %s; per kernel: counter, dispatches, last value, mean\n"
                    % (tag, "bench.py --solo --steps 6" if prefix == "pmc" else "tools/msm_bench.py"))
            for k, cs in agg.items():
                for c, v in sorted(cs.items()):
                    f.write("%-30s %-22s %4d %16.1f %16.1f\n" % (k[:30], c, len(v), v[-1], sum(v) / len(v)))
                    tail = v[-8:] if len(v) >= 8 else v      # steady state: skip one-time launches
                    if c == "FETCH_SIZE":
                        fetch[k] = sum(tail) / len(tail)
                    if c == "WRITE_SIZE":
                        write[k] = sum(tail) / len(tail)
                    if c == "SQ_INSTS_VALU":
                        valu[k] = sum(tail) / len(tail)
traffic = {"_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), the gfx950 correction of "
                    "MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B fabric requests as 64 B.  Separate --pmc passes "
                    "(tools/profile_bench.sh %s: bench.py --solo and tools/msm_bench.py), mean over the last <= 8 dispatches of "
                    "each kernel.  Kernels that run more than once per batch (k_static_accumulate: the group launch and the "
                    "re-check of failed groups) show the mean of their launches." % tag}
units = 8192
try:
    units = int(json.loads(open(os.path.join(out, "%s_solo.json" % tag)).readline()).get("batch", 8192))
except Exception:
    pass
traffic["_units_per_launch"] = units      # transactions per device batch of the profiled `bench.py --solo` run
for k in sorted(set(fetch) | set(write)):
    traffic[k] = int(1024 * (2 * fetch.get(k, 0.0) + write.get(k, 0.0)))
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
v = {"_note": "SQ_INSTS_VALU per launch (wave instructions; x 64 lanes for thread instructions), rocprofv3 --pmc pass of "
              "tools/profile_bench.sh %s, mean over the last <= 8 dispatches of each kernel" % tag}
v["_units_per_launch"] = units
v.update({k: int(x) for k, x in sorted(valu.items())})
json.dump(v, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1, sort_keys=True)
SETUP_KERNELS = {"k_tbl_multiples", "k_tbl_base", "k_from_uniform", "k_decompress", "k_spin", "__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned"}
# the side paths: SQ_INSTS_VALU per launch of every kernel of the prover / the 1032-constraint program / a transaction call
for prefix, what in (("pmcprover", "tools/prover_profile.py (2048 cloak proofs per call)"),
                     ("pmcproverprog", "tools/prover_program_profile.py (1024 proofs of the 1032-constraint program per call)"),
                     ("pmctx", "tools/tx_call_profile.py (8192 distinct serialized transactions per call)")):
    files = glob.glob(os.path.join(go, "%s_%s_SQ_WAVE_CYCLES" % (prefix, tag), "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        agg.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    side = {"_note": "SQ_INSTS_VALU per launch (wave instructions), rocprofv3 --pmc pass of tools/profile_bench.sh %s on %s: mean over "
                     "the last <= 8 dispatches of each kernel (kernels launched with several sizes per call show the mean of them); "
                     "_per_call = every dispatch of the run summed / the calls the script makes" % (tag, what)}
    with open(os.path.join(out, "%s_%s_SQ_WAVE_CYCLES.txt" % (tag, prefix)), "w") as f:
        f.write("# rocprofv3 --pmc (tools/profile_bench.sh %s), %s; per kernel: counter, dispatches, last value, mean\n" % (tag, what))
        total = calls_total = 0.0
        for k, cs in agg.items():
            for c, v in sorted(cs.items()):
                f.write("%-30s %-22s %5d %16.1f %16.1f\n" % (k[:30], c, len(v), v[-1], sum(v) / len(v)))
                if c == "SQ_INSTS_VALU":
                    tail = v[-8:] if len(v) >= 8 else v
                    side[k] = int(sum(tail) / len(tail))
                    total += sum(v)
                    if k not in SETUP_KERNELS:
                        calls_total += sum(v)
    side["_all_dispatches_valu"] = int(total)
    # the scripts say how many full calls of how many statements they made (PMC_META): instructions per full call
    try:
        import re
        meta = re.search(r"PMC_META full_calls=(\d+) batch=(\d+)", open(os.path.join(out, "%s_%s_output.txt" % (tag, prefix[3:]))).read())
        calls, batch = int(meta.group(1)), int(meta.group(2))
        side["_batch"] = batch
        side["_per_call_valu"] = int(calls_total / (calls + 8.0 / batch))      # (without the one-time kernels: tables, generators, probes)
    except Exception:
        pass
    json.dump(side, open(os.path.join(out, "pmc_valu_%s.json" % prefix[3:]), "w"), indent=1, sort_keys=True)
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag) or f.startswith("pmc_")))
# the side paths' raw summaries (prover, the 1032-constraint program, serialized transactions: kernel stats, trace summary, script
# output, their counter pass) go one level down, so that profiles/ itself holds what DESIGN.md's headline numbers quote (VERDICT r05
# item 9: at most 40 files at the top level); the tables made from them (pmc_valu_*.json) stay where bench.py reads them
side_dir = os.path.join(out, "%s_side_paths" % tag)
os.makedirs(side_dir, exist_ok=True)
for f in sorted(os.listdir(out)):
    if any(f.startswith("%s_%s" % (tag, k)) for k in ("prover", "proverprog", "tx_", "pmcprover", "pmcproverprog", "pmctx")) and os.path.isfile(os.path.join(out, f)):
        os.replace(os.path.join(out, f), os.path.join(side_dir, f))
print("side paths ->", os.path.relpath(side_dir, ROOT), sorted(os.listdir(side_dir)))
