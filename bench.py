#!/usr/bin/env python3
"""bench.py -- batch verification throughput of ZkVM cloak transactions on MI355X.

One "step" = one complete `r1cs::Verifier::verify` of every transaction of one batch, on the
device, from inputs resident in HBM: Merlin transcript replay, verification scalars
(inner-product-argument s vector, constraint flattening, g_i / h_i), decompression of the proof
points, the multiscalar multiplication per transaction and the ristretto identity test -> accept
bitmap (copied to the host).

  --config 2 (default)  BASELINE.json configs[1]: 1024 2-in/2-out cloak transactions per GPU -- the
                        1024 DISTINCT real proofs of tests/golden/cloak_2x2_1024.bin (oracle prover,
                        committed as data), ~1.5 % corrupted (n = 256 multipliers, k = 8, m = 8: 35
                        proof-specific points + 514 shared generators per transaction).
                        zkgpu_cloak_verify_submit_dev / zkgpu_verify_wait, several batches in flight.
  --config 4            BASELINE.json configs[3]: 8192 x N mixed-arity transactions (shapes 1x1, 1x2,
                        2x2, 3x3, 4x4 drawn with a fixed seed from tests/golden/cloak_mixed.bin; at
                        N = 8 the 65 536 of the config), cut into N contiguous shards balanced by the
                        number of multiscalar-multiplication terms (zkgpu_shard_cuts); each rank keeps
                        its shard resident in HBM (zkgpu_txblock), verifies it shape-grouped with
                        batches in flight (zkgpu_verifier_verify_block) and the per-shard accept bitmaps
                        are all-gathered over RCCL behind the C ABI (zkgpu_comm_allgather_bitmap);
                        every rank checks the bitmap of the WHOLE batch.

Launch:  python bench.py [--gpus N --steps K --warmup W] [--config 4]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
                --master-port P bench.py --gpus N --steps K --warmup W [--config 4]
Both forms work for N > 1: started WITHOUT a launcher (WORLD_SIZE unset) `--gpus N` makes this process spawn N fresh
rank processes itself (zkvm_amd/launch.py) before anything in it touches the GPU, relay rank 0's line and exit non-zero
if any rank did.  One process per GPU; shards are independent (weak scaling); the only collective on the data path is
the all-gather of the accept bitmaps (ncclAllGather behind zkgpu_comm_allgather_bitmap -- the ONLY RCCL communicator of
a rank; barriers, the timing reduction and the id broadcast run over gloo on the host).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the dominant kernel of the step -- by
summed solo kernel time -- against the HBM roofline the north-star names, the whole step against
it, and the integer-ALU figure that actually binds), "setup" (one-time costs that `value` does not
contain: table build, table bytes), "cpu_baseline" (the CPU oracle's full verifier -- a port of the
reference's algorithm; the reference itself is not mounted -- on this box's host cores), and in
config 2 "msm_boundary", "per_tx_checks", "host_memory", "prover", "msm_2p20" (BASELINE configs[2],
checked against tests/golden/msm_2p20.json).
"""
from __future__ import annotations

import os

# Each batch in flight has a stream of its own for its latency-bound kernels; the HIP runtime maps
# streams onto 4 hardware queues by default, which would serialise those streams again.  Must be set
# before the runtime initialises (i.e. before torch is imported): this file is the process's entry
# point, so it is.
# (ranks made to SHARE one GPU -- the two-rank test mode -- share its queue slots too: 24 each is 48 on one device, which
# then time-slices them: 0.44-0.7 M tx/s against 2.1 M with 8 each, measured)
# (N > 1, one rank per GPU: the control plane -- barrier, timing reduction, the 128-byte id broadcast -- runs over gloo on
# the host, so the ONLY RCCL communicator of a rank is the library's own (zkgpu_comm): the process holds the streams of one
# verifier and one communicator, as at N = 1, and asks for the same number of queues.  Until round 3 torch's "nccl" group
# was a second communicator per rank and the bench asked for 16 to leave it room.)
# (18 = what zkgpu_runtime_hint recommends -- the library never exports it itself; here torch may be the first HIP user: the runtime keeps that
# many queues per stream priority, the verifier's low- and default-priority streams come on top, and from 25 in all the
# device stops running them side by side in one process out of four: DESIGN.md sec 5.1, profiles/archive/r04v / r04w)
_HWQ_PRESET = "GPU_MAX_HW_QUEUES" in os.environ
# per-kernel timing (zkgpu_profile_*), the HBM copy kernel and the mode switches of the sweeps are hooks of the library
# (include/zkgpu_hooks.h), not exports: they answer only to a process that asks for them before it loads the library
os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")
if os.environ.get("ZKGPU_BENCH_SHARE_GPU"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
else:
    # the library does not edit the environment: it recommends (zkgpu_runtime_hint), the host exports -- here, before torch or
    # anything else in this process has made a HIP call
    # (torch FIRST: it brings its own copy of the HIP runtime library, and the two must resolve to ONE runtime in the process --
    # libzkgpu.so loaded before torch binds /opt/rocm's, torch then loads its own beside it, and the second one finds no
    # device.  Importing torch does not start the runtime: the hint below still comes in time, and says so if it did not.)
    import torch as _torch  # noqa: F401
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from zkvm_amd import runtime_hint as _runtime_hint
    _HWQ_HINT = _runtime_hint()
    if _HWQ_HINT[0] == 2:
        # (under `rocprofv3 --pmc` the profiler's preloaded library has started the runtime before this line: the process runs
        # on the runtime's default queues, the library reports it -- queue_info -- and the line says so; results are unaffected)
        print("bench.py: the HIP runtime had started before the queue hint could be applied (a profiler's preload?)", file=_sys.stderr)
# stdout carries exactly ONE line, the JSON record: whatever libraries print there (RCCL's version banner, gloo's
# connection chatter) is sent to stderr instead
_JSON_OUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)

import argparse
import collections
import hashlib
import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

L = 2**252 + 27742317777372353535851937790883648493
SEED = 0x5A6B564D  # "ZkVM"
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s spec
MAD_PEAK_GOPS = 1024 * 64 * 2.4 / 4.088      # v_mad_u64_u32: 4.09 cycles per wave-instruction per SIMD (profiles/r05_valu_op_rates.txt)
VALU_PEAK_GINST = 1024 * 2.4 / 4 * 64        # 1024 SIMDs x 2.4 GHz / 4 cycles per wave instruction x 64 lanes
N_SIMD, CLOCK_GHZ = 1024, 2.4                  # 256 CUs x 4 SIMDs; the clock tools/ubench/valu_ops.hip prices its cycles at
BAD_POINT = bytes.fromhex("01" + "00" * 31)
SHAPE_TERMS = {}                             # (n_in, n_out) -> (n_dyn, n_static), filled from the library
LIBRARY_MERGE = 10240                        # the LIBRARY's default merge target for tickets (session.hpp zkgpu_verifier::merge_target; tests/test_host_logic.py
                                             # holds the two equal): bench.py does not set one unless --merge is given


def emit(record) -> None:
    _JSON_OUT.write(json.dumps(record) + "\n")
    _JSON_OUT.flush()


def shake(tag: bytes, n: int) -> bytes:
    return hashlib.shake_256(SEED.to_bytes(4, "little") + tag).digest(n)


def bitmap_of(bits) -> bytes:
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i // 8] |= 1 << (i % 8)
    return bytes(out)


def bits_of(bm: bytes, n: int):
    return [(bm[i // 8] >> (i % 8)) & 1 for i in range(n)]


def usable_cores(omp_threads: int) -> int:
    """Threads the process may really run: min(OpenMP default, affinity mask, cgroup CPU quota)."""
    n = min(omp_threads, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, n)


def shape_terms(lib, n_in: int, n_out: int):
    """(dynamic terms, static terms) of the verification MSM of a shape: 11 + m + 2k and 2 + 2 pn."""
    key = (n_in, n_out)
    if key not in SHAPE_TERMS:
        total = int(lib.zkgpu_cloak_msm_terms(n_in, n_out))
        m = 2 * (n_in + n_out)
        # total = 11 + m + 2k + 2 + 2 * 2^k: solve for k
        k = next(k for k in range(1, 20) if 13 + m + 2 * k + 2 * (1 << k) == total)
        SHAPE_TERMS[key] = (11 + m + 2 * k, 2 + 2 * (1 << k))
    return SHAPE_TERMS[key]


def algorithmic_bytes(lib, shapes) -> int:
    """64 B per proof-specific term (scalar + compressed point) + 32 B per generator scalar."""
    tot = 0
    for s in shapes:
        nd, ns = shape_terms(lib, *s)
        tot += 64 * nd + 32 * ns
    return tot


# ---- workloads ---------------------------------------------------------------------------------
def workload_2x2(batch: int, rank: int, bad_every: int = 64, step: int = 0):
    """configs[1], one step: tests/gpu_util.py benched_step (shared with the -m gpu test of this very arrangement)"""
    from gpu_util import benched_step
    return benched_step(batch, rank, bad_every, step)


def profile_lanes(ctxs):
    prof = {}
    for c in ctxs:
        for k, v in c.profile_read().items():
            a = prof.get(k, (0, 0.0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
    return prof


def pmc_tables():
    """profiles/pmc_traffic.json (+ pmc_valu.json): HBM bytes and VALU instructions per launch of each kernel,
    from the separate rocprofv3 --pmc passes of tools/profile_bench.sh (None when absent)."""
    out = {}
    for name in ("pmc_traffic", "pmc_valu"):
        path = os.path.join(ROOT, "profiles", name + ".json")
        try:
            out[name] = json.load(open(path))
        except Exception:
            out[name] = None
    return out


def side_leg_counters(ctx, call, call_s, pmc_name):
    """One more call of a side leg with the library's per-launch HIP events on: the sum of its kernels' durations against
    the call's own time (how much of the call the device is busy -- kernels of one call run one after the other on its
    stream), and, with the committed SQ_INSTS_VALU pass of the same workload (profiles/<pmc_name>.json, tools/profile_bench.sh),
    the wave instructions of the call and the fraction of the chip's issue peak they make over the call's time."""
    ctx.profile_reset()
    ctx.profile(True)
    call()
    ctx.profile(False)
    prof = {k: v for k, v in ctx.profile_read().items() if v[0]}
    kernel_ms = sum(v[1] for v in prof.values())
    out = {"kernel_sum_ms": round(kernel_ms, 3), "call_ms": round(call_s * 1e3, 3), "kernel_sum_over_call": round(kernel_ms / (call_s * 1e3), 4),
           "launches": int(sum(v[0] for v in prof.values())),
           "top_kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:6]}}
    try:
        tbl = json.load(open(os.path.join(ROOT, "profiles", pmc_name + ".json")))
        missing = [k for k in prof if k not in tbl]
        insts = sum(v[0] * tbl[k] for k, v in prof.items() if k in tbl)
        out.update(valu_wave_instructions=int(insts), valu_issue_frac=round(insts / (call_s * VALU_PEAK_GINST * 1e9 / 64), 4),
                   valu_kernels_without_counter=missing or None, valu_source="profiles/%s.json (SQ_INSTS_VALU per launch x launches of this call)" % pmc_name)
    except Exception as e:                                      # noqa: BLE001
        out.update(valu_wave_instructions=None, valu_issue_frac=None, valu_source="no committed counter table (%s)" % type(e).__name__)
    return out


def prover_leg_counters(ctx, call, call_s, pmc_name, batch, pmc_batch):
    """side_leg_counters for a prover call, which the library cuts into slices that are in flight together: the kernel sum is
    taken as timed (kernels of different slices overlap, so it may exceed the call's time: `kernel_sum_over_call` > 1 then
    means overlap, not error); the wave instructions come from ONE MORE call forced into a single slice
    (zkgpu_set_prover_mode 17), whose launches are what the committed SQ_INSTS_VALU table (collected on unsliced calls of
    pmc_batch statements) describes -- scaled by batch / pmc_batch, the work being linear in the statements."""
    out = side_leg_counters(ctx, call, call_s, pmc_name)
    try:
        tbl = json.load(open(os.path.join(ROOT, "profiles", pmc_name + ".json")))
        ctx.set_prover_mode(17)
        ctx.profile_reset(); ctx.profile(True)
        call()
        ctx.profile(False)
        prof = {k: v for k, v in ctx.profile_read().items() if v[0]}
        if tbl.get("_per_call_valu"):      # every dispatch of the profiled (unsliced) calls summed, per full call of _batch statements
            insts = tbl["_per_call_valu"] * batch / float(tbl["_batch"])
        else:
            insts = sum(v[0] * tbl[k] for k, v in prof.items() if k in tbl) * batch / pmc_batch
        out.update(valu_wave_instructions=int(insts), valu_issue_frac=round(insts / (call_s * VALU_PEAK_GINST * 1e9 / 64), 4),
                   valu_kernels_without_counter=[k for k in prof if k not in tbl] or None, single_slice_kernel_sum_ms=round(sum(v[1] for v in prof.values()), 3),
                   valu_source="profiles/%s.json (SQ_INSTS_VALU per launch of an unsliced call of %d statements x its launches x %d / %d)" % (pmc_name, pmc_batch, batch, pmc_batch))
    except Exception as e:                                      # noqa: BLE001
        out.update(valu_wave_instructions=None, valu_issue_frac=None, valu_source="no committed counter table (%s)" % type(e).__name__)
    finally:
        ctx.set_prover_mode(0)
    return out


def valu_roofline(kernel, launch_ms, valu_tbl, units_per_launch):
    """The roofline that BINDS ("report HBM fraction as mandated + integer-multiply rate; say plainly which
    one binds").  From the committed SQ_INSTS_VALU of the kernel (vector wave-instructions per launch,
    profiles/pmc_valu.json) and its opcode mix (profiles/valu_mix.json: tools/isa_mix.py on the compiler's assembly x the
    per-opcode issue costs measured by tools/ubench/valu_ops.hip): the multiply-adds per second against the chip's
    multiply-add peak, and `mix_bound_ms` = the time the launch's instructions need at their own mix-weighted issue cost on
    every SIMD -- the bound no scheduling can beat -- against the measured duration."""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    except Exception:                                           # noqa: BLE001
        return None
    if kernel not in mix or kernel not in valu_tbl:
        return None
    m = mix[kernel]
    scale = units_per_launch / float(valu_tbl.get("_units_per_launch", units_per_launch))
    insts = valu_tbl[kernel] * scale                            # wave instructions of one launch
    cyc_mad = mix["_rates"].get("v_mad_u64_u32", 4.64)
    peak_gmad = N_SIMD * 64 * CLOCK_GHZ / cyc_mad               # lane multiply-adds per ns, chip-wide
    mads = insts * m["mad_frac"] * 64
    bound_ms = insts * m["cpi_mix"] / (N_SIMD * CLOCK_GHZ * 1e9) * 1e3
    return {"kernel": kernel, "bound": "valu-int", "valu_wave_instructions_per_launch": int(insts), "mad_per_launch": int(mads),
            "achieved_Gmad_s": round(mads / (launch_ms * 1e-3) / 1e9, 1), "peak_Gmad_s": round(peak_gmad, 1),
            "frac": round(mads / (launch_ms * 1e-3) / 1e9 / peak_gmad, 4), "mad_share": m["mad_frac"], "cpi_mix": m["cpi_mix"],
            "mix_bound_ms": round(bound_ms, 4), "launch_ms": round(launch_ms, 4), "mix_frac": round(bound_ms / launch_ms, 4),
            "unit": "G multiply-adds/s (v_mad_u64_u32, per lane)",
            "note": "mix_bound_ms = wave instructions x mix-weighted cycles per instruction / (%d SIMDs x %.1f GHz): what the launch's own "
                    "instruction stream costs at the measured issue rate of each opcode; mix_frac = that / measured duration -- the "
                    "fraction of the binding roofline.  frac = multiply-adds alone against the multiply-add peak (%.2f cycles each)"
                    % (N_SIMD, CLOCK_GHZ, cyc_mad)}


def roofline_object(solo, launches_per_step, in_flight_ms, alg_bytes_step, units_step, ms_per_step, table_bytes, note, counters=True):
    """solo: kernel -> mean ms alone on the chip (HIP events, measured live after the timed region);
    launches_per_step: kernel -> launches per step; dominant = largest summed solo time per step.
    counters=False: the committed PMC tables (collected on uniform 2-in/2-out launches) do not describe this workload:
    the traffic / VALU fields are null."""
    per_step = {k: solo[k] * launches_per_step.get(k, 1.0) for k in solo}
    dom = max(per_step, key=per_step.get)
    pmc = pmc_tables() if counters else {"pmc_traffic": None, "pmc_valu": None}
    traffic_tbl, valu_tbl = pmc["pmc_traffic"], pmc["pmc_valu"]
    # the dominant kernel processes every unit of the step in its launches
    alg_per_launch = alg_bytes_step / max(launches_per_step.get(dom, 1.0), 1e-9)
    achieved = alg_per_launch / (solo[dom] * 1e-3) / 1e9
    step_traffic = step_valu = None
    # the counters were collected for launches of `_units_per_launch` transactions: scaled to this run's device batch
    if traffic_tbl:
        scale = units_step / float(traffic_tbl.get("_units_per_launch", units_step))
        step_traffic = int(scale * sum(traffic_tbl.get(k, 0) * launches_per_step.get(k, 1.0) for k in solo))
    if valu_tbl:
        scale = units_step / float(valu_tbl.get("_units_per_launch", units_step))
        step_valu = int(scale * sum(valu_tbl.get(k, 0) * launches_per_step.get(k, 1.0) for k in solo))
    step = {"algorithmic_bytes": int(alg_bytes_step), "achieved": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9, 3),
            "frac": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": step_traffic,
            "traffic_over_algorithmic": round(step_traffic / alg_bytes_step, 1) if step_traffic else None,
            "hbm_busy_frac": round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if step_traffic else None,
            "valu_wave_instructions": step_valu,
            "valu_issue_frac": round(step_valu * 64 / (ms_per_step * 1e-3) / 1e9 / VALU_PEAK_GINST, 4) if step_valu else None,
            "solo_kernel_ms_sum": round(sum(per_step.values()), 4)}
    valu = valu_roofline(dom, solo[dom], valu_tbl, units_step / max(launches_per_step.get(dom, 1.0), 1e-9)) if valu_tbl else None
    by_kernel = None
    if valu_tbl:
        by_kernel = {}
        for k in sorted(per_step, key=per_step.get, reverse=True)[:8]:
            v = valu_roofline(k, solo[k], valu_tbl, units_step / max(launches_per_step.get(k, 1.0), 1e-9))
            if v:
                by_kernel[k] = {q: v[q] for q in ("mix_frac", "mix_bound_ms", "frac", "cpi_mix", "mad_share")}
    if valu:
        valu["by_kernel"] = by_kernel
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "valu": valu,
            "traffic": (int((traffic_tbl or {}).get(dom) * units_step / max(launches_per_step.get(dom, 1.0), 1e-9) /
                            float((traffic_tbl or {}).get("_units_per_launch", units_step / max(launches_per_step.get(dom, 1.0), 1e-9))))
                        if (traffic_tbl or {}).get(dom) else None),
            "algorithmic_bytes_per_launch": int(alg_per_launch),
            "units_per_launch": round(units_step / max(launches_per_step.get(dom, 1.0), 1e-9), 1),
            "avg_launch_ms": round(solo[dom], 4), "launches_per_device_batch": round(launches_per_step.get(dom, 1.0), 2),
            "launches_per_1024_tx_step": round(launches_per_step.get(dom, 1.0) * 1024.0 / max(units_step, 1), 3),
            "avg_launch_ms_in_flight": round(in_flight_ms[dom], 4) if dom in in_flight_ms else None,
            "dominant_by": "largest summed solo duration per step (HIP events around each kernel alone on the chip, same "
                           "process, after the timed region; `bench.py --solo` under rocprofv3 --stats gives the same averages)",
            "step": step,
            "binding_resource": "integer VALU (v_mad_u64_u32 field arithmetic) and the depth of the per-batch kernel DAG; "
                                "HBM moves a fraction of a percent of its peak algorithmically.  The generator tables (%.1f GB, "
                                "one random 96-B row per mixed addition) are what `traffic` mostly is." % (table_bytes / 1e9),
            "note": note}


# ---- cpu baseline ------------------------------------------------------------------------------
def cpu_baseline(txs, r_bytes, gpu_bits, budget_s=10.0):
    """Time the CPU oracle's full verifier (kind "port": transcript replay + scalars + MSM with the same radix-2^51
    field and Straus/Pippenger split as the reference's dalek back end; the Rust reference itself is not mounted)
    on a bounded sample of the same workload, and compare every accept bit with the GPU's."""
    from gpu_util import oracle_block_bits
    from oracle import binding as oracle
    cores = usable_cores(oracle.max_threads())
    one = min(32, len(txs))
    t0 = time.perf_counter()
    acc1 = oracle_block_bits(oracle, txs[:one], r_bytes[: 64 * one], threads=1)
    t1 = time.perf_counter() - t0
    assert list(acc1) == gpu_bits[:one], "GPU accept bits differ from the CPU oracle"
    est = (one / t1) * cores * 0.9
    sample = max(one, min(len(txs), int(est * budget_s)))
    reps = 0
    t0 = time.perf_counter()
    while True:
        acc = oracle_block_bits(oracle, txs[:sample], r_bytes[: 64 * sample], threads=cores)
        reps += 1
        if time.perf_counter() - t0 > budget_s * 0.8 or reps >= 64:
            break
    tall = time.perf_counter() - t0
    assert list(acc) == gpu_bits[:sample], "GPU accept bits differ from the CPU oracle"
    return {"value": round(sample * reps / tall, 1), "unit": "tx/s", "cores": cores, "kind": "port",
            "value_1core": round(one / t1, 2),
            "sample": "%d x the first %d transactions of rank 0's workload (oracle Verifier: transcript replay + scalars + MSM "
                      "per tx, same proof bytes and verifier randomness as the GPU step) on %d OpenMP threads = this box's "
                      "cgroup CPU quota (%.1f s); 1-core figure on %d tx (%.1f s); every accept bit compared with the GPU's"
                      % (reps, sample, cores, tall, one, t1)}


def msm_microbench(ctx, torch, dev):
    """BASELINE configs[2]: one 2^20-term MSM, 64 B/term (32 B scalar + 32 B compressed point), inputs regenerated from
    SHAKE256 and the result compared with the oracle's committed value (tests/golden/msm_2p20.json)."""
    from gpu_util import msm_2p20_inputs
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "msm_2p20.json")))
    n = gold["n"]
    sc, uniform = msm_2p20_inputs(n)
    pts = ctx.hash_to_points(uniform)
    assert hashlib.sha256(sc).hexdigest() == gold["scalars_sha256"] and hashlib.sha256(pts).hexdigest() == gold["points_sha256"]
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    d_pt = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    r0 = ctx.msm_dev(d_sc, d_pt, n)
    assert r0.hex() == gold["results"][str(n)], "2^20 MSM differs from the committed expected value"
    # the call as a caller sees it (no per-kernel events: each is two more packets in a queue of ~20 launches) ...
    iters = 5
    for _ in range(2):
        ctx.msm_dev(d_sc, d_pt, n)
    t0 = time.perf_counter()
    for _ in range(iters):
        r = ctx.msm_dev(d_sc, d_pt, n)
    dt = (time.perf_counter() - t0) / iters
    assert r == r0
    # ... and again with the library's HIP events around every launch, for the kernel table and the roofline of the dominant one
    ctx.profile_reset()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        r = ctx.msm_dev(d_sc, d_pt, n)
    dt_events = (time.perf_counter() - t0) / iters
    ctx.profile(False)
    assert r == r0
    prof = ctx.profile_read()
    kern = {k: round(v[1] / v[0], 4) for k, v in prof.items() if v[0]}
    n_win = 255 // ctx.last_window_bits() + 1
    out = {"terms": n, "pairs_per_s": round(n / dt, 1), "ms": round(dt * 1e3, 3), "window_bits": ctx.last_window_bits(),
           "ms_with_kernel_events": round(dt_events * 1e3, 3),
           "result": r.hex(), "equals_committed_expected_value": True, "kernel_ms": kern,
           "kernel_ms_sum": round(sum(kern.values()), 4),
           "streams": "decompression on a stream of its own beside the digit sort (explicit fork / join events); the call's "
                      "time against kernel_ms_sum shows whether the two overlapped on this box"}
    if kern:
        dom = max(kern, key=kern.get)                  # the dominant kernel BY MEASURED DURATION (HIP events, this run)
        dom_ms = kern[dom]
        madds = n * n_win                              # one mixed addition per term and window (bucket accumulation)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "algorithmic_bytes_per_launch": 64 * n,
                           "achieved": round(64 * n / (dom_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(64 * n / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "avg_launch_ms": dom_ms,
                           "whole_call_GBps": round(64 * n / dt / 1e9, 2), "whole_call_frac": round(64 * n / dt / 1e9 / HBM_PEAK_GBS, 6)}
        acc_ms = kern.get("k_bucket_accumulate", 0.0)
        if acc_ms:
            out["roofline"]["bucket_accumulate_valu_int"] = {"achieved_Gmad_s": round(madds * 700 / (acc_ms * 1e-3) / 1e9, 1), "peak_Gmad_s": MAD_PEAK_GOPS,
                                                             "frac": round(madds * 700 / (acc_ms * 1e-3) / 1e9 / MAD_PEAK_GOPS, 4)}
        valu_tbl = pmc_tables()["pmc_valu"]
        if valu_tbl:
            # wave instructions of the whole call from the committed --pmc pass of tools/msm_bench.py (same pipeline, same size)
            launches = {k: v[0] / iters for k, v in prof.items() if v[0]}
            alias = {"k_decompress": ("k_decompress_pre", "k_pow22523", "k_decompress_post", "k_decompress_fused"),
                     "k_scan": ("k_scan_reduce", "k_scan_blocksums", "k_scan_apply"), "k_bin_order": ("k_bin_classes", "k_class_scan", "k_bin_order")}
            tot = 0
            for k in kern:
                for name in alias.get(k, (k,)):
                    tot += valu_tbl.get(name, 0) * (launches.get(k, 1.0) if name == k else 1.0)
            out["valu_wave_instructions"] = int(tot)
            out["valu_issue_frac"] = round(tot * 64 / dt / 1e9 / VALU_PEAK_GINST, 4)
            # the roofline that binds, per chip-filling kernel of the call: instructions x their mix-weighted issue cost against the
            # measured duration (VERDICT r05 weak 2 read k_bucket_accumulate at 0.56 of it)
            try:
                mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
            except Exception:                                   # noqa: BLE001
                mix = {}
            by = {}
            for k, ms in (("k_pow22523", None), ("k_bucket_accumulate", kern.get("k_bucket_accumulate")), ("k_bucket_reduce_quad", kern.get("k_bucket_reduce"))):
                if k in mix and k in valu_tbl:
                    bound = valu_tbl[k] * mix[k]["cpi_mix"] / (N_SIMD * CLOCK_GHZ * 1e9) * 1e3
                    by[k] = {"valu_wave_instructions": int(valu_tbl[k]), "cpi_mix": mix[k]["cpi_mix"], "mix_bound_ms": round(bound, 4),
                             "launch_ms": ms, "mix_frac": round(bound / ms, 4) if ms else None}
            out["roofline"]["valu_by_kernel"] = by
            out["roofline"]["valu_note"] = ("mix_bound_ms = SQ_INSTS_VALU (profiles/pmc_valu.json, this pipeline) x mix-weighted cycles per instruction "
                                            "(profiles/valu_mix.json) / (1024 SIMDs x 2.4 GHz); k_pow22523 runs inside the k_decompress events "
                                            "(its own duration: the rocprofv3 summary, profiles/r06_msm_kernel_trace_summary.txt)")
    return out


def prover_program_microbench(ctx, host_threads: int, batch: int = 8192):
    """BASELINE configs[4]: R1CS proving of a 1024-constraint program -- here 8 committed values, each shown to lie in
    [0, 2^64): 512 multipliers, 1032 constraints, handed over as DATA (zkgpu_r1cs_prove_batch); every proof verified
    by the device-side verifier through a plan made from the same description."""
    import random
    from gpu_util import GADGET_LABEL, describe_ranges, gadget_witness
    from zkvm_amd.native import R1csDescription
    from zkvm_amd.verifier import BulletproofGens, R1csProver, R1csVerifier
    m, n1, n, labels, cons = describe_ranges(8)
    desc = R1csDescription(GADGET_LABEL, m, n1, n, labels, cons)
    gens = BulletproofGens(ctx, 512, table_bits=16)            # 68.9 GB beside the tables of the 2x2 generators
    rng = random.Random(SEED)
    vals, givens, seeds, mult_def = [], [], [], None
    for i in range(batch):
        values = [rng.randrange(1 << 64) for _ in range(8)]
        mult_def, given = gadget_witness(3, 8, values)
        vals.append(values); givens.append(given); seeds.append(hashlib.sha256(b"bench program %d" % i).digest())
    pr = R1csProver(ctx, gens, desc, mult_def, host_threads=host_threads)
    ctx.set_prover_mode(1)                                     # the round-1 arrangement: host threads in lockstep
    pr.prove(vals[:8], givens[:8], seeds[:8])
    pr.prove(vals[:256], givens[:256], seeds[:256])
    dt_host = pr.last_call_s
    ctx.set_prover_mode(0)                                     # the whole proof on the device
    pr.prove(vals[:8], givens[:8], seeds[:8])
    coms, proofs = pr.prove(vals, givens, seeds)
    dt = pr.last_call_s
    v = R1csVerifier(ctx, gens, desc)
    bm = v.verify_gpu(batch, b"".join(coms), b"".join(proofs), len(proofs[0]), shake(b"program-r", 64 * batch))
    v.close()
    assert bm == bitmap_of([1] * batch), "a proof of the 1032-constraint program did not verify"
    for _ in range(2):                                         # best of three calls, like the cloak leg
        pr.prove(vals, givens, seeds)
        dt = min(dt, pr.last_call_s)
    slices_used = int.from_bytes(ctx.debug_read("prover_slices", 4), "little")      # (what the library did in the timed calls)
    counters = prover_leg_counters(ctx, lambda: pr.prove(vals, givens, seeds), dt, "pmc_valu_proverprog", batch, 1024)
    gens.close()
    return {"proofs_per_s": round(batch / dt, 1), "batch": batch, "ms_per_proof": round(dt / batch * 1e3, 4), "device": counters,
            "slices": slices_used,
            "constraints": len(cons), "multipliers": n, "commitments": m, "proof_bytes": len(proofs[0]), "host_threads": host_threads,
            "host_lockstep_proofs_per_s": round(256 / dt_host, 1),
            "note": "zkgpu_r1cs_prove_batch on a described constraint system (8 x 64-bit range proofs), best of 3 calls, cut by the "
                    "library into `slices` sub-batches in flight; time of the library call: transcript, TranscriptRng, witness, flattening, polynomials and inner-product folds in kernels, "
                    "Pedersen vector commitments and every L_j / R_j on the generator tables; host_lockstep = the same with "
                    "the algebra on host threads (zkgpu_set_prover_mode 1, batch 256)"}


def tx_verify_microbench(ctx, gens, host_threads: int, verifier=None):
    """Tx::verify on SERIALIZED transactions (payment subset): wire format + VM + transaction ID on
    host threads, aggregated keys, Schnorr equations and cloak proofs on the device.  Host buffers in, PCIe included.
    The terms of this leg: every transaction of a call is DISTINCT -- built by the product's builder
    (csrc/zkvm_tx_build.hpp through libzkhost, gpu_util.built_transactions: keys, anchors, recipients and nonce of its own
    around the 1024 committed cloak proofs) -- one in 64 is damaged at DRAWN positions (proof, signature, key, header), every
    repetition hands the set over in another rotation, the figure is the MEDIAN of 5 calls, and every bit of every call is
    checked against the expectation the construction gives (which tests/test_zkvm_tx.py holds against the oracle)."""
    import statistics
    import numpy as np
    from gpu_util import built_transactions
    from zkvm_amd.verifier import BlockVerifier
    t_build = time.perf_counter()
    txs8, exp8 = built_transactions(8192, call=1, bad_every=64, threads=host_threads)
    txs32, exp32 = built_transactions(32768, call=2, bad_every=64, threads=host_threads)
    build_s = time.perf_counter() - t_build
    # on the verifier of the timed steps when there is one: a process has a limited number of hardware queues (DESIGN.md
    # sec 5.1), and a second verifier made this late in the run keeps 2 of its 6 lanes
    bv = verifier if verifier is not None else BlockVerifier(ctx, gens)
    bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)               # opt-in: the format is an unpinned recollection
    lanes_info = bv.queue_info()

    def rotated(txs, exp, k):
        r = (1031 * k) % len(txs)
        t, e = txs[r:] + txs[:r], exp[r:] + exp[:r]
        return b"".join(t), np.asarray([len(x) for x in t], dtype=np.uint64), e

    def timed_calls(txs, exp, n, reps=5):
        times = []
        for k in range(reps + 2):                              # (two untimed: a lane's first batch of a new size allocates)
            blob, lens, e = rotated(txs[:n], exp[:n], k)
            t0 = time.perf_counter()
            bm, st = bv.verify_txs_packed(blob, lens, host_threads)
            dt = time.perf_counter() - t0
            assert bm == bitmap_of(e) and list(st) == [0 if x else 1 for x in e], "a verdict differs from the constructed expectation"
            if k >= 2:
                times.append(dt)
        return statistics.median(times), min(times), max(times)

    try:
        for _ in range(bv.lanes()):
            bv.verify_txs(txs8[:4096], host_threads)
        m1, lo1, hi1 = timed_calls(txs8, exp8, 1024)
        m8, lo8, hi8 = timed_calls(txs8, exp8, 8192)
        m32, lo32, hi32 = timed_calls(txs32, exp32, 32768)
        # calls in flight on the ONE verifier (zkgpu_tx_verify_submit / _wait): 1024 transactions per call, 8 calls in flight,
        # 48 calls; the engine merges what is queued into rounds
        calls = []
        for k in range(8):
            part, e = txs8[1024 * k: 1024 * (k + 1)], exp8[1024 * k: 1024 * (k + 1)]
            calls.append((b"".join(part), np.asarray([len(x) for x in part], dtype=np.uint64), bitmap_of(e)))
        import collections
        n_calls = 48

        def flight():
            q = collections.deque()
            t0 = time.perf_counter()
            for k in range(n_calls):
                if len(q) >= 8:
                    cid, want = q.popleft()
                    bm, st = bv.wait_txs(cid)
                    assert bm == want, "a verdict of a call in flight differs from the constructed expectation"
                blob, lens, want = calls[k % 8]
                q.append((bv.submit_txs_packed(blob, lens, host_threads), want))
            while q:
                cid, want = q.popleft()
                bm, st = bv.wait_txs(cid)
                assert bm == want, "a verdict of a call in flight differs from the constructed expectation"
            return time.perf_counter() - t0

        flight()                                            # untimed: the engine thread, the second round's staging areas
        rounds0 = bv.tx_stats()
        flights = sorted(flight() for _ in range(3))
        dt_flight = flights[1]
        rounds1 = bv.tx_stats()
    finally:
        if verifier is None:
            bv.close()
    return {"tx_per_s": round(1024 / m1, 1), "batch": 1024, "ms": round(m1 * 1e3, 3), "host_threads": host_threads,
            "tx_per_s_8192_per_call": round(8192 / m8, 1), "ms_8192_per_call": round(m8 * 1e3, 3), "ms_8192_min_max": [round(lo8 * 1e3, 3), round(hi8 * 1e3, 3)],
            "tx_per_s_32768_per_call": round(32768 / m32, 1), "ms_32768_per_call": round(m32 * 1e3, 3), "ms_32768_min_max": [round(lo32 * 1e3, 3), round(hi32 * 1e3, 3)],
            "in_flight": {"tx_per_s": round(1024 * n_calls / dt_flight, 1), "tx_per_s_min_max": [round(1024 * n_calls / flights[2], 1), round(1024 * n_calls / flights[0], 1)],
                          "statistic": "median of 3 passes of 48 calls (one untimed before them)", "per_call": 1024, "calls_in_flight": 8, "calls": n_calls,
                          "rounds": rounds1[0] - rounds0[0], "calls_per_round": round((rounds1[1] - rounds0[1]) / max(1, rounds1[0] - rounds0[0]), 2)},
            "statistic": "median of 5 calls (two untimed before them)", "damaged": "1 in 64, drawn positions, five kinds",
            "distinct_transactions": [8192, 32768], "build_s": round(build_s, 2),
            "tx_bytes": len(txs8[0]), "lanes": lanes_info[0], "lanes_asked": lanes_info[1],
            "note": "zkgpu_tx_verify_batch on serialized 2-in/2-out payment transactions (host memory in): 1024, 8192 and 32 768 "
                    "DISTINCT transactions per call, 1 in 64 damaged, another rotation every repetition, median of 5, every verdict "
                    "checked; in_flight: zkgpu_tx_verify_submit / _wait, 8 calls of 1024 in flight on the one verifier, merged by "
                    "its engine into rounds.  Wire format, VM, transaction IDs and signature transcripts on a staging thread + "
                    "worker pool (AVX-512 lockstep hashing); key aggregation, signature equations and cloak proofs on the device; "
                    "Python marshalling included; the format is an unpinned recollection (opt-in), never part of `value`"}


def prover_microbench(ctx, gens, host_threads: int, batch: int = 16384, ctx2=None):
    """BASELINE configs[4] (prover side): `batch` 2-in/2-out cloak proofs per call; every proof verified by the device verifier."""
    import ctypes as C
    import random
    import threading
    from zkvm_amd.verifier import CloakTx, Prover, Verifier
    rng = random.Random(SEED)
    qs, fs, seeds = [], [], []
    for i in range(batch):
        f = rng.randrange(2**250).to_bytes(32, "little")
        a, b = rng.randrange(2**40), rng.randrange(2**40)
        qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3])
        fs.append([f] * 4)
        seeds.append(hashlib.sha256(b"bench prover %d" % i).digest())
    qa = (C.c_uint64 * (4 * batch))(*[q for row in qs for q in row])
    fl, sd = b"".join(f for row in fs for f in row), b"".join(seeds)
    # the prover is mostly multiscalar multiplications ON THE GENERATORS (a third of its time is k_static_accumulate): for it the
    # widest tables pay -- 79 k proofs/s over 16-bit tables against 69 k over the 14-bit ones the library chooses for a
    # verifier (zkgpu_choose_table_bits: the knee of the VERIFIER's sweep) -- so this leg builds 16-bit tables of its own
    from zkvm_amd.verifier import BulletproofGens
    own_gens = None
    if gens.points.table_bits() != 16:
        own_gens = gens = BulletproofGens(ctx, 256, table_bits=16)
    pr = Prover(ctx, gens, host_threads=host_threads)
    ctx.set_prover_mode(1)                                     # the round-1 arrangement: host threads in lockstep
    pr.prove(2, 2, qs[:8], fs[:8], seeds[:8])
    pr.prove(2, 2, qs[:512], fs[:512], seeds[:512])
    dt_host = pr.last_call_s
    ctx.set_prover_mode(0)                                     # the whole proof on the device
    pr.prove_packed(2, 2, batch, qa, fl, sd)
    best = None
    for _ in range(3):
        com, proofs, plen = pr.prove_packed(2, 2, batch, qa, fl, sd)
        best = pr.last_call_s if best is None else min(best, pr.last_call_s)
    slices_used = int.from_bytes(ctx.debug_read("prover_slices", 4), "little")      # (what the library did in the timed calls)
    # two calls in flight: two host threads, each on a context of its own (a fork: same tables)
    second = Prover(ctx2 if ctx2 is not None else ctx.fork(), gens, host_threads=max(1, host_threads // 2))   # (forks are a limited resource)
    pr.host_threads = max(1, host_threads // 2)
    second.prove_packed(2, 2, batch, qa, fl, sd)
    rounds = 4
    def work(p):
        for _ in range(rounds):
            p.prove_packed(2, 2, batch, qa, fl, sd)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(p,)) for p in (pr, second)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt2 = time.perf_counter() - t0
    stride = 1 + 32 * (16 + 2 * 16)
    txs = [CloakTx(2, 2, com.raw[256 * i: 256 * (i + 1)], proofs.raw[stride * i: stride * i + plen]) for i in range(batch)]
    v = Verifier(ctx, gens)
    bm = v.verify_bitmap_gpu(txs, shake(b"prover-r", 64 * batch))
    v.close()
    assert bm == bitmap_of([1] * batch), "a proof of the GPU prover did not verify"
    pr.host_threads = host_threads
    counters = prover_leg_counters(ctx, lambda: pr.prove_packed(2, 2, batch, qa, fl, sd), best, "pmc_valu_prover", batch, 2048)
    if own_gens is not None:
        own_gens.close()
    return {"proofs_per_s": round(batch / best, 1), "batch": batch, "ms_per_proof": round(best / batch * 1e3, 4), "device": counters,
            "generator_table_bits": 16, "slices": slices_used,
            "two_calls_in_flight_proofs_per_s": round(2 * rounds * batch / dt2, 1),
            "host_threads": host_threads, "host_lockstep_proofs_per_s": round(512 / dt_host, 1),
            "note": "zkgpu_cloak_prove_batch on contiguous inputs, time of the library call: the whole proof on the device "
                    "(k_pv_* kernels, one workgroup per proof), all multiscalar multiplications on the generator tables; host "
                    "threads only derive the blinding factors and the gadget's witness queue; the library cuts the call into "
                    "`slices` sub-batches in flight on streams of their own (round 5).  two_calls_in_flight: two host "
                    "threads, each calling on a context of its own (the host share of one call beside the device share of the "
                    "other); host_lockstep = zkgpu_set_prover_mode 1 (batch 512); every proof verified by the device-side verifier"}


# ---- distributed plumbing ------------------------------------------------------------------------
class World:
    """One process per GPU.  Control plane (barrier, max over ranks, the 128-byte communicator id, the per-rank records)
    over gloo on the host; the data path's one collective is the library's (zkgpu_comm, RCCL)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.bringup = None                                  # (N > 1: how the communicator's bring-up went -- in the line)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (self.world, args.gpus))
        if args.gpus > 1 and self.world == 1:               # (main() spawns the ranks before it gets here)
            raise SystemExit("--gpus %d needs %d rank processes" % (args.gpus, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (libzkgpu has no CPU fallback)")
        # ZKGPU_BENCH_SHARE_GPU=1 (exercising the N > 1 code path on a 1-GPU box): every rank uses device 0 and the
        # bitmaps travel over gloo from host memory instead of RCCL (which refuses two ranks on one device)
        self.share_gpu = os.environ.get("ZKGPU_BENCH_SHARE_GPU") == "1"
        if not self.share_gpu and self.world > torch.cuda.device_count():
            raise SystemExit("%d ranks but %d GPUs visible (ZKGPU_BENCH_SHARE_GPU=1 puts every rank on device 0: a rehearsal, "
                             "not a measurement)" % (self.world, torch.cuda.device_count()))
        self.local = 0 if self.share_gpu else local
        if self.share_gpu and self.world > 2 and args.table_bits < 0:
            # more than two ranks on ONE device (the N = 8 rehearsal): every rank builds its own tables and workspaces in the
            # one device's memory -- 8 x (15 GB of tables + ~16 GB of lanes) of config 4 do not fit 288 GB (the first
            # rehearsal ended in HSA_STATUS_ERROR_OUT_OF_RESOURCES).  Narrow tables: a correctness run, not a measurement.
            args.table_bits = 12
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.coll_dev = torch.device("cpu")
        self.no_rccl = None                                  # (set by make_exchange when the communicator cannot be made: the reason)
        self.dist = None
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # the whole run of a rank is bounded too (zkvm_amd/bringup.py: Watchdog): RCCL's all-gather in the timed region
            # cannot be cancelled either
            from zkvm_amd import bringup
            self.run_watchdog = bringup.Watchdog(float(getattr(args, "run_timeout", 700.0)), "this rank's run (a collective in the timed region?)",
                                                 code=4, rank=self.rank).start()
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(minutes=20))
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, mine):
        """every rank's object, in rank order, on every rank (collective)"""
        if not self.dist:
            return [mine]
        out = [None] * self.world
        self.dist.all_gather_object(out, mine)
        return out

    def broadcast_bytes(self, b: bytes, n: int) -> bytes:
        if not self.dist:
            return b
        t = self.torch.frombuffer(bytearray(b if self.rank == 0 else bytes(n)), dtype=self.torch.uint8)
        self.dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def describe_ranks(self):
        """per rank: the device it runs on, and the collective library's version -- so that a multi-GPU record says by
        itself what it ran on (collective: every rank calls this)"""
        torch = self.torch
        p = torch.cuda.get_device_properties(self.local)
        mine = {"rank": self.rank, "device": self.local, "name": p.name, "gcn_arch": getattr(p, "gcnArchName", ""),
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
                "hbm_GiB": round(p.total_memory / 2**30, 1), "pid": os.getpid(),
                "launched_by": os.environ.get("ZKGPU_LAUNCHED_BY", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "the caller")}
        try:
            mine["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:                                  # noqa: BLE001
            mine["rccl_version"] = "unknown (%s)" % type(e).__name__
        return self.gather_objects(mine)

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.run_watchdog.cancel()


def rccl_roll_call(W, comm):
    """What RCCL itself saw: every rank contributes its rank number through the library's communicator (a real
    ncclAllGather when world > 1) and reports how many distinct ranks came back and the communicator's own world size."""
    if comm is None:
        return None
    got = comm.allgather(W.rank.to_bytes(4, "little"))
    seen = sorted({int.from_bytes(got[4 * i: 4 * i + 4], "little") for i in range(W.world)})
    assert seen == list(range(W.world)), "the all-gather returned ranks %s in a world of %d" % (seen, W.world)
    return {"rccl_ranks_seen": len(seen), "zkgpu_comm_world": int(comm.ctx.lib.zkgpu_comm_world(comm.h))}


def common_line(args, W, value, elapsed, data, config):
    return {"metric": "ZkVM tx verifications/sec (batch)", "value": round(value, 1), "unit": "tx/s", "n_gpus": W.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limb pairs of radix-2^51 (v_mad_u64_u32)", "data": data, "config": config}


# ---- config 2: the headline --------------------------------------------------------------------
class StepSet:
    """One step's batch resident in HBM: commitments, proofs, verifier randomness (torch tensors) and the accept bitmap
    the construction implies."""
    __slots__ = ("com", "proofs", "r", "want", "bits", "txs", "r_bytes", "host")


def make_exchange(W, ctx, cuts, always_comm=False, comm=None):
    """The one exchange step of the sharded path -- rank r's accept bitmap of [cuts[r], cuts[r+1]) in, the bitmap of the
    whole batch out on every rank -- through the C ABI (zkgpu_comm_allgather_bitmap: ncclAllGather over xGMI) in both
    configs; gloo only when several ranks share one GPU (ZKGPU_BENCH_SHARE_GPU, the 1-GPU rehearsal: RCCL refuses two
    ranks on one device).  -> (exchange(local_bitmap, status) -> whole bitmap, close())"""
    if W.world == 1 and (not always_comm or W.no_rccl):
        def alone(local, status=0):
            if status:
                raise RuntimeError("verification failed with status %d" % status)
            return local
        return alone, (lambda: None), (W.no_rccl or "none (one rank)"), None
    parts = [(cuts[i], cuts[i + 1]) for i in range(W.world)]
    # (ZKGPU_BENCH_TRY_RCCL=1 with ranks sharing a GPU: RCCL is tried all the same -- it refuses, which rehearses the fallback below)
    if (W.share_gpu and os.environ.get("ZKGPU_BENCH_TRY_RCCL") != "1") or W.no_rccl:
        from zkvm_amd.sharded import gather_bitmaps
        return (lambda local, status=0: gather_bitmaps(parts, local, status != 0, W.dist, None)), (lambda: None), (W.no_rccl or "gloo (ranks share one GPU)"), None
    from zkvm_amd.native import Comm
    own = comm is None
    if own:
        # The communicator, and one real all-gather through it, before anything is timed -- and BOUNDED (zkvm_amd/bringup.py):
        # ncclCommInitRank cannot be cancelled from inside a process, so (1) every rank first tries the bring-up in a CHILD
        # process it can kill (a stall or a refusal there costs nothing: the ranks agree over gloo, the bitmaps then travel
        # over gloo, and the line says so and why -- the run still measures the verification); (2) the bring-up in the rank
        # itself, after every child has succeeded, runs under a watchdog that ends the rank with exit code 3 and the reason
        # on standard error, so that the launcher ends the others: never a silent wait for the driver's limit.
        from zkvm_amd import bringup
        T = float(getattr(W.args, "comm_timeout", 120.0))
        errs = [None] * W.world
        if W.world > 1 and os.environ.get("ZKGPU_BENCH_COMM_PROBE", "1") != "0":
            t0 = time.perf_counter()
            errs = bringup.probe(W.rank, W.world, W.local, T, W.broadcast_bytes, W.gather_objects)
            W.bringup = {"probe_s": round(time.perf_counter() - t0, 2), "probe": "child process per rank: zkgpu_comm_create + one all-gather"}
        if not any(errs):
            err = None
            with bringup.Watchdog(T, "zkgpu_comm_create (ncclCommInitRank) + the first all-gather", code=3, rank=W.rank):
                uid = W.broadcast_bytes(Comm.unique_id() if W.rank == 0 else b"", 128)
                try:
                    comm = Comm(ctx, W.rank, W.world, uid)
                    got = comm.allgather(W.rank.to_bytes(4, "little"))
                    if [int.from_bytes(got[4 * i: 4 * i + 4], "little") for i in range(W.world)] != list(range(W.world)):
                        raise RuntimeError("the first all-gather returned the wrong ranks")
                except Exception as e:                              # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, str(e)[:300])
                errs = W.gather_objects(err)
        if any(errs):
            if comm is not None:
                try:
                    comm.close()
                except Exception:                               # noqa: BLE001
                    pass
            bad = next(i for i, e in enumerate(errs) if e)
            W.no_rccl = "gloo -- RCCL could not be brought up on rank %d (%s)" % (bad, errs[bad])
            print("bench.py: " + W.no_rccl, file=sys.stderr)
            return make_exchange(W, ctx, cuts, always_comm, None)
    ex = (lambda local, status=0: comm.allgather_bitmap(cuts, local, status)), (comm.close if own else (lambda: None)), "ncclAllGather via zkgpu_comm_allgather_bitmap"
    return ex + (comm,)


def run_config2(args, W):
    torch, dev, rank, world = W.torch, W.dev, W.rank, W.world
    from gpu_util import benched_randomness
    from zkvm_amd import Context
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens, CloakTx, Verifier
    ctx = Context(W.local)
    lib = ctx.lib
    batch = args.batch
    host_threads = max(1, usable_cores(os.cpu_count() or 1) // max(1, world))
    t0 = time.perf_counter()
    gens = BulletproofGens(ctx, 256, table_bits=args.table_bits)
    table_s = time.perf_counter() - t0
    table_bytes = int(lib.zkgpu_pointset_table_bytes(gens.points.h))
    args.table_bits = gens.points.table_bits()           # (-1: what the library chose)

    def to_dev(b, dtype=torch.uint8):
        return torch.frombuffer(bytearray(b), dtype=dtype).to(dev)

    # Every step is a batch of its own (workload_2x2): distinct proofs order, distinct corruptions, distinct verifier
    # randomness.  The ring is as long as the longest run of this process, so no batch is ever submitted twice while
    # anything that met it could still sit in a cache: 1.35 MB per step in HBM.
    steady_steps = 0 if (args.lean or args.solo or args.no_steady or world > 1) else 200
    lanes_planned = args.inflight if args.tickets > 0 else 1
    n_sets = max(args.steps + max(args.warmup, 1, lanes_planned * (args.merge // batch)), steady_steps, 2 * args.merge // batch + 2, 24)
    sets = []
    for sidx in range(n_sets):
        txs, expected = workload_2x2(batch, rank, args.bad_every, sidx)
        r_bytes = benched_randomness(rank, sidx, batch)
        S = StepSet()
        S.com, S.proofs, S.r = to_dev(b"".join(t[2] for t in txs)), to_dev(b"".join(t[3] for t in txs)), to_dev(r_bytes)
        S.bits, S.want = expected, bitmap_of(expected)
        S.txs, S.r_bytes = (txs, r_bytes) if sidx == 0 else (None, None)
        # the same step in HOST memory for the host_memory leg (the first 64 steps: 1.4 MB each)
        S.host = (b"".join(t[2] for t in txs), b"".join(t[3] for t in txs), r_bytes) if sidx < 64 and rank == 0 else None
        sets.append(S)
    txs, r_bytes = sets[0].txs, sets[0].r_bytes
    n_in, n_out = txs[0][0], txs[0][1]
    n_dyn, n_static = shape_terms(lib, n_in, n_out)
    ctxs_txs = [CloakTx(*t) for t in txs]
    proof_len = len(txs[0][3])
    nbytes = (batch + 7) // 8
    cuts = [batch * i for i in range(world + 1)]
    exchange, close_exchange, exchange_name, comm0 = make_exchange(W, ctx, cuts)
    exchanges, shared_comm = {1: (exchange, close_exchange, exchange_name, comm0)}, [comm0]

    # `--inflight M`: M device batches in flight.  Each has its own forked context (workspace + a light stream for its
    # latency-bound kernels); the chip-filling kernels of all of them go first-in first-out through the parent's
    # shared streams (zkgpu_ctx_fork).  Every step is one complete, independent verification of a whole batch.
    # the modes of include/zkgpu_hooks.h are touched only when a flag asks for something else than the library's own choice:
    # the timed path is the path a process without hooks runs
    if args.group != 16:
        ctx.set_group_size(args.group)           # forks inherit it
    if args.transcript_mode:
        ctx.set_transcript_mode(args.transcript_mode)
    if args.locate_mode:
        ctx.set_locate_mode(args.locate_mode)
    if args.locate_parts:
        ctx.set_locate_parts(args.locate_parts)
    if args.tail_mode:
        ctx.set_tail_mode(args.tail_mode)
    if args.horner_mode:
        ctx.set_horner_mode(args.horner_mode)
    # (with tickets the plain forks serve the single-rank side legs only: at N > 1 they would be four more streams beside the
    # verifier's lanes and the communicator's, on a device whose queue slots are limited -- DESIGN.md sec 5.1)
    ctxs = [ctx] + [ctx.fork() for _ in range((min(max(1, args.inflight), 10) if args.tickets <= 0 else (5 if world == 1 else 1)) - 1)]
    gv = Verifier(ctx, gens)
    torch.cuda.synchronize()
    host_time = {"submit": 0.0, "n": 0}
    whole_last = [None]

    # N > 1: every step's accept bitmap is exchanged (zkgpu_comm_allgather_bitmap: ncclAllGather over xGMI) and the whole
    # bitmap checked on every rank -- the bitmaps of the steps that one merged device batch finished together travel in ONE
    # collective (a status word + 128 bytes per step and rank: latency-bound either way), not in one collective per step
    gather_every = max(1, args.merge // batch) if args.tickets > 0 else 1
    pend_x = []

    def flush_exchange():
        if not pend_x:
            return
        g = len(pend_x)
        cuts_g = [batch * g * i for i in range(world + 1)]
        ex = exchanges.get(g)
        if ex is None:
            ex = exchanges[g] = make_exchange(W, ctx, cuts_g, comm=shared_comm[0])
        whole = ex[0](b"".join(bm for bm, _ in pend_x), 0)
        mine = whole[rank * nbytes * g:(rank + 1) * nbytes * g]
        assert mine == b"".join(bm for bm, _ in pend_x)
        whole_last[0] = whole
        pend_x.clear()

    def checked(bm, j, gather=True):
        assert bm == sets[j].want, "accept bitmap of step set %d differs from the constructed expectation" % j
        if gather and world > 1:
            pend_x.append((bm, j))
            if len(pend_x) >= gather_every:
                flush_exchange()
        return bm

    def run_steps(n, base=0, submit=None, gather=True, lanes=None):
        # plain contexts (no tickets): step base + i on lane i mod depth; gather=False: the rank-0-only extra legs
        lanes = lanes or ctxs
        depth = len(lanes)
        pend = collections.deque()
        bm = None
        for i in range(n):
            c = lanes[i % depth]
            if len(pend) >= depth:
                cc, j = pend.popleft()
                bm = checked(cc.verify_wait(), j, gather)
            j = (base + i) % n_sets
            ts = time.perf_counter()
            if submit is not None:
                submit(c, j)
            else:
                gv.submit_packed_gpu_dev(n_in, n_out, batch, sets[j].com, sets[j].proofs, proof_len, sets[j].r, ctx=c)
            host_time["submit"] += time.perf_counter() - ts
            host_time["n"] += 1
            pend.append((c, j))
        while pend:
            cc, j = pend.popleft()
            bm = checked(cc.verify_wait(), j, gather)
        flush_exchange()
        return bm

    bv = None
    if args.tickets > 0:
        bv = BlockVerifier(ctx, gens, batches_in_flight=args.inflight)
        if args.merge_given:
            bv.set_merge(args.merge)
        if args.group != 16:
            for i in range(bv.lanes()):
                bv.lane(i).set_group_size(args.group)
        # setup, untimed, as the table build is: every lane's workspace for the largest device batch the merge policy can form
        # (1.5 x the target: zkgpu.h "Tickets"), by the call the header recommends to an integrator for exactly this
        bv.reserve(n_in, n_out, 3 * args.merge // 2)

    def run_tickets(n, base=0, depth=None, gather=True):
        # `--tickets D`: up to D batches in flight as tickets; the verifier merges them into device batches of --merge
        # transactions.  The batches that arrive together are queued by ONE call.
        depth = depth or args.tickets
        q, bm = collections.deque(), None
        first = min(n, depth)
        idx = [(base + i) % n_sets for i in range(first)]
        ts = time.perf_counter()
        tk = bv.submit_many_dev(n_in, n_out, batch, [sets[j].com for j in idx], [sets[j].proofs for j in idx], proof_len,
                                [sets[j].r for j in idx])
        host_time["submit"] += time.perf_counter() - ts
        host_time["n"] += first
        q.extend(zip(tk, idx))
        for i in range(first, n):
            if len(q) >= depth:
                t, j = q.popleft()
                bm = checked(bv.wait(t), j, gather)
            j = (base + i) % n_sets
            ts = time.perf_counter()
            q.append((bv.submit_dev(n_in, n_out, batch, sets[j].com, sets[j].proofs, proof_len, sets[j].r), j))
            host_time["submit"] += time.perf_counter() - ts
            host_time["n"] += 1
        while q:                                              # every step's bitmap is checked and exchanged, the drained ones too
            t, j = q.popleft()
            bm = checked(bv.wait(t), j, gather)
        flush_exchange()
        return bm

    def run_tickets_host(n, base=0, depth=None):
        # the same arrangement fed from HOST memory (zkgpu_verifier_submit / _submit_many): what a caller of Tx::verify has.
        # Step sets 0..63 (their host copies), cycled.
        depth = depth or args.tickets
        q, bm = collections.deque(), None
        first = min(n, depth)
        idx = [(base + i) % 64 for i in range(first)]
        tk = bv.submit_many(n_in, n_out, batch, [sets[j].host[0] for j in idx], [sets[j].host[1] for j in idx], proof_len,
                            [sets[j].host[2] for j in idx])
        q.extend(zip(tk, idx))
        for i in range(first, n):
            if len(q) >= depth:
                t, j = q.popleft()
                bm = checked(bv.wait(t), j, False)
            j = (base + i) % 64
            q.append((bv.submit(n_in, n_out, batch, sets[j].host[0], sets[j].host[1], proof_len, sets[j].host[2]), j))
        while q:
            t, j = q.popleft()
            bm = checked(bv.wait(t), j, False)
        return bm

    # what the device executes are merged batches of --merge transactions: the per-kernel figures (solo pass, PMC passes)
    # are taken on such a batch, made of `rep` DISTINCT steps side by side, exactly what the verifier's merge produces
    rep = max(1, args.merge // batch) if bv is not None else 1
    dev_batch = rep * batch

    def merged_inputs(first_set):
        js = [(first_set + q) % n_sets for q in range(rep)]
        if rep == 1:
            S = sets[js[0]]
            return S.com, S.proofs, S.r, S.want
        out = (torch.cat([sets[j].com for j in js]), torch.cat([sets[j].proofs for j in js]), torch.cat([sets[j].r for j in js]),
               bitmap_of([b for j in js for b in sets[j].bits]))
        torch.cuda.synchronize()      # (torch.cat runs on torch's stream; the library's streams do not wait for it)
        return out

    def solo_pass(reps=5):
        ctx.profile_reset()
        ctx.set_serial(True)
        ctx.profile(True)
        for k in range(reps):
            d_c, d_p, d_r, want = merged_inputs(k * rep)      # a different device batch every repetition
            gv.submit_packed_gpu_dev(n_in, n_out, dev_batch, d_c, d_p, proof_len, d_r, ctx=ctx)
            assert ctx.verify_wait() == want
        ctx.profile(False)
        ctx.set_serial(False)
        prof = ctx.profile_read()
        return {k: v[1] / v[0] for k, v in prof.items() if v[0]}, {k: v[0] / reps for k, v in prof.items() if v[0]}

    if args.solo:      # what tools/profile_bench.sh runs under rocprofv3 --stats / --pmc: every kernel alone on the chip
        solo_pass(2)
        solo, launches = solo_pass(max(args.steps, 5))
        if rank == 0:
            emit({"solo_kernel_ms": {k: round(v, 4) for k, v in sorted(solo.items())}, "launches_per_step": launches, "batch": dev_batch,
                  "inputs": "%d distinct steps per device batch, a different device batch every repetition" % rep})
        gv.close()
        if bv is not None:
            bv.close()
        for c in ctxs[1:]:
            c.close()
        gens.close()
        ctx.close()
        return

    timed = run_tickets if bv is not None else run_steps
    n_warm = max(args.warmup, 1)
    # Workspace priming (setup, untimed, as the table build is): every lane executes one device batch of the size the
    # timed steps will give it, so that its workspace exists -- the library never allocates in steady state, but a lane's
    # FIRST batch of a new size does (hipMalloc synchronises the device).  Then the W warm-up steps proper.
    n_prime = (bv.lanes() * rep if bv is not None else len(ctxs))
    timed(n_prime, base=args.steps, gather=False)
    timed(n_warm, base=args.steps)                            # warm-up on sets the timed steps do not use
    # The timed region runs WITHOUT the per-launch HIP events (zkgpu_profile_*, a hook): it is the path an integrator's process
    # runs.  The in-flight kernel durations come from a second, untimed pass over the same steps afterwards (rank 0).
    # (ZKGPU_BENCH_INFLIGHT_EVENTS=1 puts the events back into the timed region, as until round 5.)
    prof_ctxs = [bv.lane(i) for i in range(bv.lanes())] if bv is not None else ctxs[:1]
    inflight_events = os.environ.get("ZKGPU_BENCH_INFLIGHT_EVENTS") == "1"
    for c in prof_ctxs:
        c.profile_reset()
        c.profile(inflight_events)
    W.barrier()
    t0 = time.perf_counter()
    host_time.update(submit=0.0, n=0)
    bm = timed(args.steps, base=0)
    own_elapsed = time.perf_counter() - t0                     # this rank's own steps, before it waits for the others
    W.barrier()
    elapsed = time.perf_counter() - t0
    submit_ms = host_time["submit"] / max(host_time["n"], 1) * 1e3
    for c in prof_ctxs:
        c.profile(False)
    elapsed = W.max_over_ranks(elapsed)
    ranks_info = W.describe_ranks()
    per_rank = W.gather_objects({"rank": rank, "tx_per_s": round(batch * args.steps / own_elapsed, 1), "ms": round(own_elapsed * 1e3, 3)})
    roll_call = rccl_roll_call(W, shared_comm[0])

    if rank == 0:
        if not inflight_events and world == 1:
            for c in prof_ctxs:
                c.profile_reset()
                c.profile(True)
            timed(args.steps, base=0, gather=False)                # the same steps once more, untimed, with events around every launch
            for c in prof_ctxs:
                c.profile(False)
        prof = profile_lanes(prof_ctxs)
        in_flight_ms = {k: v[1] / v[0] for k, v in prof.items() if v[0]}
        solo, launches = solo_pass()
        ms_per_step = elapsed / args.steps * 1e3
        alg_step = algorithmic_bytes(lib, [(n_in, n_out)] * batch)
        alg_dev = alg_step * (dev_batch // batch)                 # per device batch (the unit the kernels are launched for)
        ms_per_dev_batch = ms_per_step * (dev_batch // batch)
        line = common_line(args, W, batch * world * args.steps / elapsed, elapsed,
                           "synthetic: 1024 distinct real R1CS proofs of the 2-in/2-out cloak statement (committed fixture, oracle "
                           "prover); every step is a batch of its own -- the fixture rotated, its own ~1.5% corruptions, its own "
                           "64 bytes of verifier randomness per transaction (SHAKE256 of rank and step): no two batches in flight, "
                           "and no two merged into one device batch, share scalars or table rows",
                           {"workload": "BASELINE configs[1]: batch of %d 2-in/2-out cloak tx per GPU, complete r1cs::Verifier::verify "
                                        "on the device from commitments + R1CSProof bytes + verifier randomness resident in HBM: "
                                        "Merlin replay, verification scalars (IPA s vector, constraint flattening), decompression, "
                                        "the %d-term mega_check MSM (n=256, k=8, m=8; %d terms on shared generators), identity "
                                        "test -> accept bitmap" % (batch, n_dyn + n_static, n_static),
                            "tx_per_gpu": batch, "terms_per_tx": n_dyn + n_static, "generator_table_bits": args.table_bits,
                            "calls_in_flight": min(args.tickets, args.steps) if bv is not None else len(ctxs), "group_size": args.group,
                            "merged_device_batches": ({"transactions": args.merge, "lanes": bv.lanes(), "source": "--merge" if args.merge_given else "library default",
                                                       "policy": "what is queued of one shape leaves in round(queued / target) device batches of equal size"}
                                                      if bv is not None else None),
                            "events_in_timed_region": inflight_events,
                            "distinct_step_inputs": n_sets, "exchange": exchange_name, "steps_per_exchange": gather_every if world > 1 else None,
                            "ranks": ranks_info, "per_rank": per_rank, "rccl": roll_call,
                            "control_plane": "gloo (host)" if world > 1 else None, "bringup": W.bringup,
                            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0")),
                            "hw_queues_set_by": ("caller" if _HWQ_PRESET else "nobody: the HIP runtime had started before bench.py could export the hint (runtime default)"
                                                 if "GPU_MAX_HW_QUEUES" not in os.environ else "bench.py on zkgpu_runtime_hint's advice, before the HIP runtime started"),
                            "parallelism": "tx-sharded x%d, RCCL all-gather of accept bitmaps" % world})
        line["roofline"] = roofline_object(solo, launches, in_flight_ms, alg_dev, dev_batch, ms_per_dev_batch, table_bytes,
                                           "algorithmic bytes per launch = %d B per transaction (64 B x %d proof-specific terms + 32 B x "
                                           "%d generator scalars) x the %d transactions a launch processes (%d batches of %d merged by the "
                                           "verifier); `step` figures are per device batch" % (alg_step // batch, n_dyn, n_static, dev_batch, dev_batch // batch, batch))
        line["setup"] = {"table_build_ms": round(table_s * 1e3, 1), "table_bytes": table_bytes,
                         "note": "one-time per generator set (generators + fixed-base tables, zkgpu_pointset_build_tables); not in `value`"}
        line["kernel_ms_in_flight"] = {k: round(x, 4) for k, x in sorted(in_flight_ms.items())}
        line["kernel_ms_solo"] = {k: round(x, 4) for k, x in sorted(solo.items())}
        line["host_submit_ms_per_step"] = round(submit_ms, 4)
        if bv is not None and not args.lean:
            # ONE batch of 1024 with nothing else in flight (configs[1] read literally): submit, wait, repeat
            lat = []
            for k in range(12):
                j = (args.steps + k) % n_sets
                t1 = time.perf_counter()
                t = bv.submit_dev(n_in, n_out, batch, sets[j].com, sets[j].proofs, proof_len, sets[j].r)
                got = bv.wait(t)
                lat.append(time.perf_counter() - t1)
                assert got == sets[j].want
            lat.sort()
            line["latency_one_batch_ms"] = round(lat[len(lat) // 2] * 1e3, 4)
            line["latency_one_batch"] = {"ms_median": round(lat[len(lat) // 2] * 1e3, 4), "ms_min": round(lat[0] * 1e3, 4),
                                         "tx_per_s": round(batch / lat[len(lat) // 2], 1),
                                         "note": "one %d-transaction batch, nothing else in flight: zkgpu_verifier_submit_dev + "
                                                 "zkgpu_verifier_wait, 12 distinct batches" % batch}
        if bv is not None and not args.lean and world == 1:
            # `value` against the step count (it used to peak where --steps was a multiple of merge / 1024):
            # the same timed arrangement -- one submit_many of everything, then the waits -- for other run lengths, best of 3
            by_steps = {}
            for s_n in (16, 20, 24, 37):
                best = None
                for rep_i in range(3):
                    t1 = time.perf_counter()
                    run_tickets(s_n, base=(7 * rep_i) % n_sets, gather=False)
                    dt = time.perf_counter() - t1
                    best = dt if best is None or dt < best else best
                by_steps[str(s_n)] = round(batch * s_n / best, 1)
            line["value_by_steps"] = dict(by_steps, note="tx/s of a run of that many 1024-transaction steps, everything queued by one call "
                                          "(best of 3): 16 / 20 / 24 / 37 steps leave as 2 x 8, 2 x 10, 2 x 12, 10 + 9 + 9 + 9 tickets per device "
                                          "batch.  What still differs is the ~1.1 ms tail behind the LAST batch (Horner chains, verdicts), paid "
                                          "once per run and amortised over more transactions in a longer one")
        if bv is not None and steady_steps:
            # the same arrangement under sustained load: 200 steps, up to 64 tickets in flight (never `value` unless the
            # run itself is that long)
            run_tickets(24, base=0, depth=64, gather=False)
            t1 = time.perf_counter()
            run_tickets(steady_steps, base=0, depth=64, gather=False)
            dt = time.perf_counter() - t1
            line["steady_state"] = {"tx_per_s": round(batch * steady_steps / dt, 1), "ms_per_step": round(dt / steady_steps * 1e3, 4),
                                    "steps": steady_steps, "tickets_in_flight": 64,
                                    "note": "same verifier, same merge target, 200 distinct steps with up to 64 tickets in flight"}
        host_tickets = None
        if bv is not None and not args.lean:
            # tickets from host memory: the timed arrangement again (same --steps, same tickets in flight, same merge target),
            # every step's bytes handed over from host memory; then 200 steps of it
            run_tickets_host(max(bv.lanes() * rep, n_warm), base=32)                    # staging areas are made here
            t1 = time.perf_counter()
            run_tickets_host(args.steps, base=0)
            dt = time.perf_counter() - t1
            host_tickets = {"tx_per_s": round(batch * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps}
            if steady_steps:
                t1 = time.perf_counter()
                run_tickets_host(steady_steps, base=0, depth=64)
                dt = time.perf_counter() - t1
                host_tickets["steady_tx_per_s"] = round(batch * steady_steps / dt, 1)
        if not args.lean:
            line["hbm_copy"] = hbm_copy_leg(ctx)
            line["cpu2"] = ("omitted: BASELINE.md's second CPU line (libsodium naive sum of crypto_scalarmult_ristretto255) needs "
                            "libsodium on the GPU box; it exists only in the build container (/opt/conda/lib), where the golden vectors "
                            "were made, and nothing outside /root/repo travels")
            ps = gens.points
            # the multiscalar-multiplication boundary alone: scalars prepared beforehand by the product's host verifier
            hv = Verifier(ctx, gens, host_threads=host_threads)
            t0 = time.perf_counter()
            prep = hv.prepare(ctxs_txs, r_bytes)
            prep_s = time.perf_counter() - t0
            d_dyn_sc, d_dyn_pt, d_st_sc = to_dev(prep["dyn_sc"]), to_dev(prep["dyn_pt"]), to_dev(prep["st_sc"])
            d_st_idx = torch.tensor(prep["st_idx"], dtype=torch.int32, device=dev)
            d_dyn_off = torch.tensor(prep["dyn_off"], dtype=torch.int64, device=dev)
            d_st_off = torch.tensor(prep["st_off"], dtype=torch.int64, device=dev)

            def submit_msm_only(c, j):
                c.verify_batch_ps_submit_dev(ps, batch, d_dyn_sc, d_dyn_pt, d_dyn_off, batch * n_dyn,
                                             d_st_sc, d_st_idx, d_st_off, batch * n_static)
            assert all(int(x) for x in prep["wellformed"])

            def run_set0(n, submit):
                depth, bm0 = len(ctxs), None
                for i in range(n + depth):
                    c = ctxs[i % depth]
                    if i >= depth:
                        bm0 = c.verify_wait()
                        assert bm0 == sets[0].want
                    if i < n:
                        submit(c, 0)
                return bm0
            run_set0(len(ctxs), submit_msm_only)
            t0 = time.perf_counter()
            run_set0(args.steps, submit_msm_only)
            msm_only_s = (time.perf_counter() - t0) / args.steps
            # proof bytes -> accept bits with the verifier head on host threads
            t0 = time.perf_counter()
            bm_e2e = hv.verify_bitmap(ctxs_txs, r_bytes)
            e2e_s = time.perf_counter() - t0
            assert bm_e2e == sets[0].want
            # the device path fed from HOST memory (PCIe copies + python marshalling included)
            packed_com = b"".join(t[2] for t in txs)
            packed_proofs = b"".join(t[3] for t in txs)
            n_e2e = 6 * len(ctxs)

            def submit_host(c, j):
                gv.submit_packed_gpu(n_in, n_out, batch, packed_com, packed_proofs, proof_len, r_bytes, ctx=c)
            run_set0(len(ctxs), submit_host)
            t0 = time.perf_counter()
            run_set0(n_e2e, submit_host)
            e2e_gpu_s = (time.perf_counter() - t0) / n_e2e
            # the same complete verification with every transaction checked on its own (no group checks)
            for c in ctxs:
                c.set_group_size(1)
            run_steps(len(ctxs), base=0, gather=False)
            t0 = time.perf_counter()
            run_steps(args.steps, base=0, gather=False)
            per_tx_s = (time.perf_counter() - t0) / args.steps
            for c in ctxs:
                c.set_group_size(args.group)
            line["per_tx_checks"] = {"tx_per_s": round(batch / per_tx_s, 1), "ms_per_step": round(per_tx_s * 1e3, 4),
                                     "note": "the same steps on plain contexts (%d batches of %d in flight, no merging) with "
                                             "zkgpu_set_group_size(1): every transaction's MSM on its own" % (len(ctxs), batch)}
            line["msm_boundary"] = {"tx_per_s": round(batch / msm_only_s, 1), "ms_per_step": round(msm_only_s * 1e3, 4),
                                    "note": "zkgpu_verify_batch_ps_submit_dev alone: decompress + MSM + identity test on scalars "
                                            "prepared beforehand (the argument list of dalek's mega_check resident in HBM)"}
            line["host_memory"] = {"gpu_resident_tx_per_s": host_tickets["tx_per_s"] if host_tickets else round(batch / e2e_gpu_s, 1),
                                   "tickets": host_tickets,
                                   "per_context_submit_tx_per_s": round(batch / e2e_gpu_s, 1),
                                   "host_prepared_tx_per_s": round(batch / e2e_s, 1), "host_threads": host_threads,
                                   "host_prepare_ms_per_batch": round(prep_s * 1e3, 2),
                                   "note": "proof bytes in host memory -> accept bits, PCIe copies and python marshalling included.  "
                                           "gpu_resident (= tickets.tx_per_s): the timed arrangement of `value` -- same steps, same tickets in "
                                           "flight, same merge target -- with every step handed over from host memory "
                                           "(zkgpu_verifier_submit_many for the first wave, zkgpu_verifier_submit after it: pinned staging, "
                                           "three copies per device batch on the verifier's copy stream); tickets.steady_tx_per_s: 200 steps "
                                           "of it.  per_context_submit: zkgpu_cloak_verify_submit on plain contexts (no merging; round 3's "
                                           "figure).  host_prepared: zkgpu_cloak_verify_batch (verifier head on %d host threads).  None is "
                                           "`value`." % host_threads}
            hv.close()
            if world == 1 and not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline(txs, r_bytes, bits_of(sets[0].want, batch))
            if world == 1 and not args.no_sweep:
                line["setup"]["table_bits_sweep"] = table_bits_sweep(args, ctx, ctxs, sets, n_in, n_out, proof_len, batch, gens, torch)
            if world == 1 and not args.no_msm:
                # side legs: a failure of one of them is reported in its field, it does not take the headline line with it
                # (a wrong RESULT in a leg is an AssertionError and does)
                # the headline's verifier and all but one of the plain contexts are finished with: their streams go back to the
                # runtime BEFORE the legs make theirs (a process has a limited number of hardware queues, DESIGN.md sec 5.1: a
                # verifier made beside them keeps 2 of its 6 lanes)
                if bv is not None:
                    bv.close()
                    bv = None
                gv.close()
                gv = None
                for c in ctxs[2:]:
                    c.close()
                del ctxs[2:]
                for key, leg in (("tx_verify", lambda: tx_verify_microbench(ctx, gens, host_threads)),
                                 ("prover", lambda: prover_microbench(ctx, gens, host_threads, ctx2=ctxs[1])),
                                 ("prover_1024_constraints", lambda: prover_program_microbench(ctx, host_threads)),
                                 ("msm_2p20", lambda: msm_microbench(ctx, torch, dev))):
                    try:
                        line[key] = leg()
                    except AssertionError:
                        raise
                    except Exception as e:                     # noqa: BLE001
                        line[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        emit(line)
    close_exchange()
    W.close()
    if bv is not None:
        bv.close()
    if gv is not None:
        gv.close()
    for c in ctxs[1:]:
        c.close()
    gens.close()
    ctx.close()


def hbm_copy_leg(ctx):
    """'measure achievable HBM with a copy kernel and report both' -- the library's own streaming
    copy kernel (k_hbm_copy: one 16-byte vector per lane, one workgroup per 4 KiB) over 2 x 1 GiB, bytes read + bytes written
    per second.  The microarchitecture guide quotes 6.29 TB/s for the same kind of kernel."""
    gbs = ctx.measure_hbm_copy(1 << 30, 10)
    return {"measured_copy_GBps": round(gbs, 1), "spec_GBps": HBM_PEAK_GBS, "measured_over_spec": round(gbs / HBM_PEAK_GBS, 4),
            "guide_copy_GBps": 6290.0,
            "note": "zkgpu_measure_hbm_copy: 1 GiB read + 1 GiB written per launch, best of 10 launches (HIP events); `roofline.peak` "
                    "stays the 8 TB/s spec the north-star names; divide `roofline.frac` by measured_over_spec for the fraction of "
                    "what a copy achieves"}


def table_bits_sweep(args, ctx, lanes, sets, n_in, n_out, proof_len, batch, gens16, torch):
    """Throughput of the same verification over generator tables of 13 .. 16 bits (what zkgpu_pointset_build_tables(.., 0)
    chooses between by capacity and free HBM): device batches of 8192 distinct transactions, 5 in flight on plain contexts."""
    from zkvm_amd.verifier import BulletproofGens, Verifier
    rep, out = 8, {}
    n_big = min(6, len(sets) // rep)
    big = []
    for k in range(n_big):
        js = range(k * rep, (k + 1) * rep)
        big.append((torch.cat([sets[j].com for j in js]), torch.cat([sets[j].proofs for j in js]), torch.cat([sets[j].r for j in js]),
                    bitmap_of([b for j in js for b in sets[j].bits])))
    torch.cuda.synchronize()          # (torch.cat runs on torch's stream; the library's streams do not wait for it)
    for w in (13, 14, 15, 16):
        t0 = time.perf_counter()
        g = gens16 if w == args.table_bits else BulletproofGens(ctx, 256, table_bits=w)
        build_s = time.perf_counter() - t0
        v = Verifier(ctx, g)
        try:
            def run(n):
                depth = len(lanes)
                for i in range(n + depth):
                    c = lanes[i % depth]
                    if i >= depth:
                        assert c.verify_wait() == big[(i - depth) % n_big][3]
                    if i < n:
                        d_c, d_p, d_r, _ = big[i % n_big]
                        v.submit_packed_gpu_dev(n_in, n_out, rep * batch, d_c, d_p, proof_len, d_r, ctx=c)
            run(len(lanes))
            t0 = time.perf_counter()
            n = 15
            run(n)
            dt = time.perf_counter() - t0
            out[str(w)] = {"tx_per_s": round(n * rep * batch / dt, 1), "table_bytes": int(ctx.lib.zkgpu_pointset_table_bytes(g.points.h)),
                           "table_build_ms": None if g is gens16 else round(build_s * 1e3, 1)}
        finally:
            v.close()
            if g is not gens16:
                g.close()
    out["note"] = "device batches of %d distinct transactions, %d in flight on plain contexts (no tickets), groups of %d" % (rep * batch, len(lanes), args.group)
    return out


# ---- config 4: mixed-arity blocks sharded over the node -------------------------------------------
def run_config4(args, W):
    torch, rank, world = W.torch, W.rank, W.world
    from gpu_util import mixed_block
    from zkvm_amd import Context, ZkGpuError
    from zkvm_amd.native import shard_cuts
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens, CloakTx
    ctx = Context(W.local)
    lib = ctx.lib
    per_gpu = args.batch if args.batch != 1024 else 8192
    total = per_gpu * world
    # the whole batch is known to every rank (a block as every node of the network sees it); deterministic
    txs = mixed_block(total, seed=SEED)
    r_bytes = shake(b"config4 verifier-r", 64 * total)
    shapes = [(t[0], t[1]) for t in txs]
    cuts = shard_cuts(shapes, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    t0 = time.perf_counter()
    gens = BulletproofGens(ctx, 512, table_bits=args.table_bits)
    table_s = time.perf_counter() - t0
    table_bytes = int(lib.zkgpu_pointset_table_bytes(gens.points.h))
    args.table_bits = gens.points.table_bits()
    bv = BlockVerifier(ctx, gens, batches_in_flight=args.inflight, chunk=args.chunk)
    for i in range(bv.lanes()):
        bv.lane(i).set_group_size(args.group)
        bv.lane(i).set_transcript_mode(args.transcript_mode)
        bv.lane(i).set_locate_mode(args.locate_mode)
        bv.lane(i).set_locate_parts(args.locate_parts)
        bv.lane(i).set_tail_mode(args.tail_mode)
        bv.lane(i).set_horner_mode(args.horner_mode)
    # since round 4 a block's batches are tickets: batches of one shape from the blocks in flight are merged into device batches
    # of up to `merge` transactions (8192 here).  Every lane's workspace is sized for that beforehand (zkgpu_verifier_reserve):
    # batches of five shapes and varying sizes would otherwise keep growing workspaces -- hipMalloc synchronises the device --
    # for many steps
    merge4 = args.merge if args.merge_given else 8192
    bv.set_merge(merge4)
    for shape in sorted(set(shapes[lo:hi])):
        bv.reserve(shape[0], shape[1], merge4)
    mine = [CloakTx(*t) for t in txs[lo:hi]]
    block = bv.block(mine, r_bytes[64 * lo: 64 * hi])          # this rank's shard, resident in HBM, grouped by shape
    # the exchange step: RCCL behind the C ABI (zkgpu_comm), the same function config 2 uses; gloo when several ranks share one GPU
    exchange, close_exchange, exchange_name, _comm = make_exchange(W, ctx, cuts, always_comm=True)   # (a world of one too: through RCCL)
    parts = [(cuts[i], cuts[i + 1]) for i in range(world)]

    # a step = one whole block: its batches queued (zkgpu_verifier_block_start), its verdicts waited for
    # (zkgpu_verifier_block_finish) and exchanged.  --blocks-in-flight D (default 4) queues blocks k+1 .. k+D-1 before block k is
    # waited for, as a node verifying a stream of blocks does: since round 4 the batches of one shape from the blocks in flight
    # are MERGED (they are tickets of the verifier's queue) instead of competing for lanes.  Measured on one MI355X, 8192 mixed
    # transactions per block (profiles/archive/r04p_*, r04q_*): one block at a time 2.0 M tx/s (five batches of ~1640 transactions, a
    # burst with its own head and tail every step); 2 / 3 / 4 / 6 / 8 in flight at merge 8192: 2.6 / 2.8 / 2.93 / 3.0 / 3.09 M.
    # Until round 3 a second block in flight bought nothing (1.96 vs 1.97 M: it waited for lanes).
    depth = max(1, args.blocks_in_flight)

    def start():
        try:
            return bv.block_start(block), 0
        except ZkGpuError as e:
            return None, e.code

    def finish(handle):
        run, status = handle
        local = b""
        if run is not None:
            try:
                local = bv.block_finish(run)
            except ZkGpuError as e:
                status = e.code
        return exchange(local, status)

    def steps(k):
        inflight, out = collections.deque(), None
        for _ in range(k):
            inflight.append(start())
            while len(inflight) >= depth:
                out = finish(inflight.popleft())
                assert out == want, "whole-batch accept bitmap differs from the constructed expectation (rank %d)" % rank
        while inflight:
            out = finish(inflight.popleft())
            assert out == want, "whole-batch accept bitmap differs from the constructed expectation (rank %d)" % rank
        return out

    # expectation by construction: the corruptions of mixed_block reject, everything else is a valid proof;
    # the -m gpu test compares the same construction bit by bit with the oracle
    expected = [0 if i % 61 == 3 else 1 for i in range(total)]
    want = bitmap_of(expected)
    whole = steps(max(args.warmup, 1))
    W.barrier()
    t0 = time.perf_counter()
    whole = steps(args.steps)
    own_elapsed = time.perf_counter() - t0
    W.barrier()
    elapsed = W.max_over_ranks(time.perf_counter() - t0)
    assert whole == want
    ranks_info = W.describe_ranks()
    per_rank = W.gather_objects({"rank": rank, "tx_per_s": round((hi - lo) * args.steps / own_elapsed, 1), "ms": round(own_elapsed * 1e3, 3)})
    roll_call = rccl_roll_call(W, _comm)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        counts = {}
        for s in shapes:
            counts["%dx%d" % s] = counts.get("%dx%d" % s, 0) + 1
        shard_terms = [sum(sum(shape_terms(lib, *s)) for s in shapes[a:b]) for a, b in parts]
        alg_step = algorithmic_bytes(lib, shapes[lo:hi])
        # per-kernel durations: a second verifier with ONE lane, every kernel alone on the chip
        solo_v = BlockVerifier(ctx, gens, batches_in_flight=1, chunk=args.chunk)
        lane0 = solo_v.lane(0)
        lane0.set_group_size(args.group)
        sblock = solo_v.block(mine, r_bytes[64 * lo: 64 * hi])
        solo_v.verify_block(sblock)
        lane0.profile_reset()
        lane0.set_serial(True)
        lane0.profile(True)
        reps = 2
        for _ in range(reps):
            assert solo_v.verify_block(sblock) == bitmap_of(expected[lo:hi])
        lane0.profile(False)
        lane0.set_serial(False)
        prof = lane0.profile_read()
        solo = {k: v[1] / v[0] for k, v in prof.items() if v[0]}
        launches = {k: v[0] / reps for k, v in prof.items() if v[0]}
        sblock.close()
        solo_v.close()
        line = common_line(args, W, total * args.steps / elapsed, elapsed,
                           "synthetic: 160 real R1CS proofs (32 of each shape, committed fixture, oracle prover) drawn with a fixed "
                           "seed, every transaction under its own verifier randomness; 1 in 61 corrupted (six kinds, every shape)",
                           {"workload": "BASELINE configs[3]: %d mixed-arity cloak tx (%d per GPU; shapes 1x1, 1x2, 2x2, 3x3, 4x4), "
                                        "sharded over %d GPU(s) by multiscalar-multiplication terms, each shard resident in HBM and "
                                        "verified shape-grouped with batches in flight, RCCL all-gather of the accept bitmaps behind "
                                        "the C ABI, whole bitmap checked on every rank" % (total, per_gpu, world),
                            "tx_total": total, "tx_per_gpu": per_gpu, "per_shape": counts,
                            "shard_tx": [b - a for a, b in parts], "shard_terms": shard_terms,
                            "generator_table_bits": args.table_bits, "gens_capacity": 512, "calls_in_flight": bv.lanes(), "blocks_in_flight": depth,
                            "chunk": args.chunk or 2048, "group_size": args.group,
                            "exchange": exchange_name, "ranks": ranks_info, "per_rank": per_rank, "rccl": roll_call,
                            "control_plane": "gloo (host)" if world > 1 else None, "bringup": W.bringup,
                            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0")),
                            "parallelism": "tx-sharded x%d" % world})
        line["roofline"] = roofline_object(solo, launches, {}, alg_step, hi - lo, ms_per_step, table_bytes,
                                           "rank 0's shard: %d transactions, %d algorithmic bytes (64 B per proof-specific term + 32 B "
                                           "per generator scalar, per shape); the committed counter tables were collected on uniform "
                                           "2-in/2-out launches and are not applied to mixed shapes (traffic / VALU fields null)"
                                           % (hi - lo, alg_step), counters=False)
        line["setup"] = {"table_build_ms": round(table_s * 1e3, 1), "table_bytes": table_bytes,
                         "note": "one table set for every shape up to 4x4 (512 + 512 generators); not in `value`"}
        line["kernel_ms_solo"] = {k: round(x, 4) for k, x in sorted(solo.items())}
        if not args.lean:
            # the same shard handed over in HOST memory (grouping, PCIe copies, python marshalling included)
            t0 = time.perf_counter()
            hb = bv.verify(mine, r_bytes[64 * lo: 64 * hi])
            host_s = time.perf_counter() - t0
            assert hb == bitmap_of(expected[lo:hi])
            line["host_memory"] = {"tx_per_s": round((hi - lo) / host_s, 1),
                                   "note": "zkgpu_verifier_verify on rank 0's shard from host memory; not `value`"}
            if world == 1 and not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline(txs[lo:hi], r_bytes[64 * lo: 64 * hi], bits_of(whole, total)[lo:hi])
        emit(line)
    close_exchange()
    W.close()
    block.close()
    bv.close()
    gens.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4),
                    help="2: BASELINE configs[1], 1024 2x2 tx per GPU (default, the headline); 4: configs[3], mixed arity, sharded")
    ap.add_argument("--batch", type=int, default=1024, help="transactions per GPU (config 4: default 8192)")
    ap.add_argument("--table-bits", type=int, default=-1, help="window width of the fixed-base generator tables (-1: the library chooses the knee of additions per term against table bytes: 14 for 514 points on an MI355X)")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="--gpus N > 1 started by this program: seconds before the launcher ends every rank and returns 124")
    ap.add_argument("--run-timeout", type=float, default=float(os.environ.get("ZKGPU_BENCH_RUN_TIMEOUT", "700")),
                    help="N > 1: seconds a rank may take for its whole run (a collective that stalls in the timed region ends the rank with "
                         "exit code 4 and the reason on standard error instead of waiting for the launcher's or the driver's limit)")
    ap.add_argument("--comm-timeout", type=float, default=float(os.environ.get("ZKGPU_BENCH_COMM_TIMEOUT", "120")),
                    help="N > 1: seconds the RCCL communicator's bring-up (probe in a child process, then ncclCommInitRank + one all-gather in the rank) may take")
    ap.add_argument("--inflight", type=int, default=0,
                    help="device batches in flight per GPU (contexts): default 5 with --tickets, 6 without, 6 for config 4")
    ap.add_argument("--group", type=int, default=16, help="transactions per group check (1 = every transaction on its own)")
    ap.add_argument("--chunk", type=int, default=0, help="config 4: transactions per batch in flight (0 = library default)")
    ap.add_argument("--blocks-in-flight", type=int, default=4, help="config 4: blocks started before the oldest is waited for; their batches of one shape are merged (default 4)")
    ap.add_argument("--bad-every", type=int, default=64, help="config 2: one transaction in this many is corrupted (0 = none)")
    ap.add_argument("--tickets", type=int, default=-1,
                    help="config 2: batches kept in flight as tickets of a zkgpu_verifier, which merges them into device batches of --merge tx (0 = plain contexts)")
    ap.add_argument("--merge", type=int, default=0, help="config 2 with --tickets: zkgpu_verifier_set_merge (0: not called -- the library's default, %d)" % LIBRARY_MERGE)
    ap.add_argument("--locate-mode", type=int, default=0, choices=(0, 1, 2, 3), help="zkgpu_set_locate_mode")
    ap.add_argument("--locate-parts", type=int, default=0, help="zkgpu_set_locate_parts (0 = the library's default)")
    ap.add_argument("--tail-mode", type=int, default=0, choices=(0, 1), help="zkgpu_set_tail_mode")
    ap.add_argument("--horner-mode", type=int, default=0, choices=(0, 1, 2), help="zkgpu_set_horner_mode")
    ap.add_argument("--transcript-mode", type=int, default=0, choices=(0, 1, 2),
                    help="zkgpu_set_transcript_mode: 0 automatic, 1 one lane per transaction, 2 one wavefront per transaction")
    ap.add_argument("--lean", action="store_true", help="the timed steps and the solo pass only (what tools/profile_bench.sh profiles)")
    ap.add_argument("--solo", action="store_true", help="config 2: only serial steps, every kernel alone on the chip (for rocprofv3 --stats)")
    ap.add_argument("--no-steady", action="store_true", help="config 2: skip the 200-step steady-state leg")
    ap.add_argument("--no-sweep", action="store_true", help="config 2: skip the generator-table width sweep")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-msm", action="store_true", help="skip the prover and 2^20 MSM legs")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.config == 2 else 20
    if args.warmup is None:
        args.warmup = 10 if args.config == 2 else 3
    if args.config == 4:
        args.tickets = 0
    # ONE arrangement whatever the flags: up to 64 batches of 1024 in flight as tickets, merged by the verifier into device
    # batches of --merge transactions on 5 lanes.  (Until round 2 a short run -- the driver's --steps 20 -- was reshaped
    # into two device batches of steps / 2 batches each; the arrangement no longer looks at --steps: a run shorter than 64
    # steps simply has fewer tickets in flight, and its last device batch is as large as what is left.)
    if args.tickets < 0:
        args.tickets = 64
    args.merge_given = args.merge > 0
    if args.merge <= 0:
        args.merge = LIBRARY_MERGE                    # (for the bench's own arithmetic: ring length, priming, the solo pass's device batch)
    if args.inflight <= 0:
        args.inflight = 5 if args.tickets > 0 else (10 if args.config == 4 and args.blocks_in_flight > 1 else 6)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher above us: this process becomes one.  It has not imported torch and never touches the GPU; the ranks
        # are fresh interpreters running this same command line (zkvm_amd/launch.py), rank 0's line is relayed as ours.
        from zkvm_amd.launch import spawn_ranks
        rc, codes = spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, out=_JSON_OUT, err=sys.stderr,
                                timeout=args.launch_timeout)
        if rc != 0:
            print("bench.py: rank exit codes %s" % codes, file=sys.stderr)
        sys.exit(rc)
    W = World(args)
    (run_config2 if args.config == 2 else run_config4)(args, W)


if __name__ == "__main__":
    main()
