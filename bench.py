#!/usr/bin/env python3
"""bench.py -- batch verification throughput of the Bulletproofs-R1CS MSM tail on MI355X.

One "step" = one pass of the hot path (zkgpu_verify_batch_ps_dev: decompress the
proof points, Pippenger MSM per transaction, identity test, accept bitmap) over
one batch of synthetic 2-in/2-out-cloak-shaped verification equations whose
inputs are already resident in HBM.  Workload = BASELINE.json configs[1]:
1024 transactions per GPU, each an MSM of 2n + 2k + m + 13 = 549 terms with
n = 256 multipliers, k = lg n = 8, m = 8 commitments; 514 terms use the shared
generators (B, B_blinding, G[0..256), H[0..256)) held in a device point set, 35
carry their own compressed points.  ~1.5 % of the transactions are corrupted so
the accept bitmap is not trivial.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
One process per GPU; shards are independent (weak scaling: 1024 tx per GPU); the
only collective is the RCCL all-gather of the per-shard accept bitmaps.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel vs
the HBM roofline the north-star names, plus the integer-ALU figure that
actually binds), "cpu_baseline" (the CPU oracle -- a port of the reference's
algorithm, the reference itself is not mounted -- on the host cores of this
box), "msm_2p20" (BASELINE configs[2] microbench, outside the timed region).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L = 2**252 + 27742317777372353535851937790883648493
SEED = 0x5A6B564D  # "ZkVM"
N_MULT, LG_N, N_COMMIT = 256, 8, 8
N_DYN = 6 + N_COMMIT + 5 + 2 * LG_N          # 35 proof-specific points
N_STATIC = 2 + 2 * N_MULT                    # 514 generator terms
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s spec
# measured v_mad_u64_u32 issue rate, profiles/r01_valu_rates.txt (Gop/s, chip-wide)
MAD_PEAK_GOPS = 33864.9
# v_mad_u64_u32 per mixed point addition: 7 field mul x 100 (field.hpp)
MADS_PER_MADD = 700


def shake(tag: bytes, n: int) -> bytes:
    return hashlib.shake_256(SEED.to_bytes(4, "little") + tag).digest(n)


def scalars_from_stream(tag: bytes, n: int) -> list:
    raw = shake(tag, 64 * n)
    return [int.from_bytes(raw[64 * i: 64 * i + 64], "little") % L for i in range(n)]


def sc_bytes(xs) -> bytes:
    return b"".join(x.to_bytes(32, "little") for x in xs)


def build_workload(ctx, batch: int, rank: int, table_bits: int = 0):
    """Synthesise `batch` verification equations with the product library only
    (generator derivation, hash-to-point and the closing point all run through
    libzkgpu; the oracle is not involved here)."""
    from zkvm_amd import PointSet
    b, bb = ctx.pedersen_gens()
    g, h = ctx.bulletproof_gens(N_MULT)
    static_points = b + bb + g + h
    ps = PointSet(ctx, static_points)
    if table_bits:
        ps.build_tables(table_bits)
    pool_n = 509
    pool = ctx.hash_to_points(shake(b"pool|%d" % rank, 64 * pool_n))
    tag = b"r%d|" % rank
    st = scalars_from_stream(tag + b"static", batch * N_STATIC)
    dy = scalars_from_stream(tag + b"dyn", batch * N_DYN)
    dyn_pts = []
    for i in range(batch):
        for j in range(N_DYN):
            k = (i * 131 + j * 17 + rank) % pool_n
            dyn_pts.append(pool[32 * k: 32 * k + 32])
    # closing point: last dynamic term of every tx is a * Q with Q = -(1/a) * (sum of the other 548 terms)
    rows_sc, rows_pt, offs = [], [], [0]
    for i in range(batch):
        rows_sc.append(sc_bytes(st[i * N_STATIC:(i + 1) * N_STATIC]) + sc_bytes(dy[i * N_DYN:(i + 1) * N_DYN - 1]))
        rows_pt.append(static_points + b"".join(dyn_pts[i * N_DYN:(i + 1) * N_DYN - 1]))
        offs.append(offs[-1] + N_STATIC + N_DYN - 1)
    partial, ok = ctx.msm_batch(b"".join(rows_sc), b"".join(rows_pt), offs)
    assert ok == bytes([0xFF] * (batch // 8)) + (bytes([(1 << (batch % 8)) - 1]) if batch % 8 else b"")
    a = [dy[(i + 1) * N_DYN - 1] or 1 for i in range(batch)]
    neg_inv = sc_bytes([(L - pow(x, -1, L)) % L for x in a])
    closing, ok2 = ctx.msm_batch(neg_inv, partial, list(range(batch + 1)))
    expected = [1] * batch
    dyn_sc = bytearray(sc_bytes(dy))
    dyn_pt = bytearray(b"".join(dyn_pts))
    st_sc = bytearray(sc_bytes(st))
    for i in range(batch):
        o = ((i + 1) * N_DYN - 1) * 32
        dyn_sc[o:o + 32] = a[i].to_bytes(32, "little")
        dyn_pt[o:o + 32] = closing[32 * i: 32 * i + 32]
    # corrupt ~1.5 %: alternate a wrong generator scalar / an undecodable proof point / a wrong proof scalar
    for c, i in enumerate(range(7, batch, 64)):
        expected[i] = 0
        if c % 3 == 0:
            o = (i * N_STATIC + 5) * 32
            st_sc[o] ^= 1
        elif c % 3 == 1:
            o = (i * N_DYN + 3) * 32
            dyn_pt[o:o + 32] = bytes.fromhex("01" + "00" * 31)
        else:
            o = (i * N_DYN + 9) * 32
            dyn_sc[o] ^= 4
    dyn_off = [i * N_DYN for i in range(batch + 1)]
    st_off = [i * N_STATIC for i in range(batch + 1)]
    return {"ps": ps, "static_points": static_points, "dyn_sc": bytes(dyn_sc), "dyn_pt": bytes(dyn_pt),
            "st_sc": bytes(st_sc), "dyn_off": dyn_off, "st_off": st_off, "expected": expected}


def bitmap_of(bits) -> bytes:
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i // 8] |= 1 << (i % 8)
    return bytes(out)


def flatten_rows(w, rows):
    """generic CSR (all points compressed) for the rows in `rows` -- what the CPU oracle consumes"""
    sc, pt, offs = [], [], [0]
    for i in rows:
        sc.append(w["st_sc"][i * N_STATIC * 32:(i + 1) * N_STATIC * 32] + w["dyn_sc"][i * N_DYN * 32:(i + 1) * N_DYN * 32])
        pt.append(w["static_points"] + w["dyn_pt"][i * N_DYN * 32:(i + 1) * N_DYN * 32])
        offs.append(offs[-1] + N_STATIC + N_DYN)
    return b"".join(sc), b"".join(pt), offs


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


def cpu_baseline(w, gpu_bitmap: bytes, batch: int):
    """Time the CPU oracle (kind "port": same radix-2^51 field and Straus/Pippenger split as the
    reference's dalek back end; the Rust reference itself is not mounted) on this box's cores,
    and check its accept bits against the GPU's on the sampled transactions."""
    from oracle import binding as oracle
    cores = usable_cores(oracle.max_threads())
    one = list(range(0, min(batch, 24)))
    sc, pt, offs = flatten_rows(w, one)
    t0 = time.perf_counter()
    bm1 = oracle.verify_batch(sc, pt, offs, threads=1)
    t1 = time.perf_counter() - t0
    rows = list(range(batch))
    sc, pt, offs = flatten_rows(w, rows)
    reps = 0
    t0 = time.perf_counter()
    while True:
        bm = oracle.verify_batch(sc, pt, offs, threads=cores)
        reps += 1
        if time.perf_counter() - t0 > 6.0 or reps >= 50:
            break
    tall = time.perf_counter() - t0
    assert bm == gpu_bitmap, "GPU accept bitmap differs from the CPU oracle"
    assert all(((bm1[i // 8] >> (i % 8)) & 1) == ((gpu_bitmap[i // 8] >> (i % 8)) & 1) for i in one)
    return {"value": round(batch * reps / tall, 1), "unit": "tx/s", "cores": cores, "kind": "port",
            "value_1core": round(len(one) / t1, 2),
            "sample": "%d x the full %d-tx batch on %d OpenMP threads = this box's cgroup CPU quota (%.1f s); 1-core figure on %d tx (%.1f s); "
                      "accept bits compared with the GPU's" % (reps, batch, cores, tall, len(one), t1)}


def msm_microbench(ctx, torch, dev):
    """BASELINE configs[2]: one 2^20-term MSM, 64 B/term (32 B scalar + 32 B compressed point)."""
    n = 1 << 20
    pts = ctx.hash_to_points(shake(b"msm2p20", 64 * n))
    g = torch.Generator(device="cpu").manual_seed(SEED)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    sc[:, 31] &= 0x0F                                    # < 2^252 < l
    d_sc = sc.to(dev)
    d_pt = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    r0 = ctx.msm_dev(d_sc, d_pt, n)
    ctx.profile_reset()
    ctx.profile(True)
    iters = 5
    t0 = time.perf_counter()
    for _ in range(iters):
        r = ctx.msm_dev(d_sc, d_pt, n)
    dt = (time.perf_counter() - t0) / iters
    ctx.profile(False)
    assert r == r0
    prof = ctx.profile_read()
    kern = {k: round(v[1] / v[0], 4) for k, v in prof.items() if v[0]}
    acc_ms = kern.get("k_bucket_accumulate", 0.0)
    return {"terms": n, "pairs_per_s": round(n / dt, 1), "ms": round(dt * 1e3, 3), "window_bits": ctx.last_window_bits(),
            "algorithmic_GBps_whole_call": round(64 * n / dt / 1e9, 2),
            "kernel_ms": kern, "result": r.hex()[:16],
            "accumulate_GBps": round(64 * n / (acc_ms * 1e-3) / 1e9, 2) if acc_ms else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="transactions per GPU")
    ap.add_argument("--table-bits", type=int, default=int(os.environ.get("ZKGPU_TABLE_BITS", "13")),
                    help="window width of the fixed-base generator tables (0 = no tables: Pippenger for every term)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ZKGPU_INFLIGHT", "3")),
                    help="independent verify calls in flight per GPU (contexts x host threads)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-msm", action="store_true", help="skip the 2^20 MSM microbench")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libzkgpu has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zkvm_amd import Context
    ctx = Context(local)
    batch = args.batch
    w = build_workload(ctx, batch, rank, args.table_bits)

    def to_dev(b, dtype=torch.uint8):
        return torch.frombuffer(bytearray(b), dtype=dtype).to(dev)

    d_dyn_sc, d_dyn_pt, d_st_sc = to_dev(w["dyn_sc"]), to_dev(w["dyn_pt"]), to_dev(w["st_sc"])
    d_dyn_off = torch.tensor(w["dyn_off"], dtype=torch.int64, device=dev)
    d_st_off = torch.tensor(w["st_off"], dtype=torch.int64, device=dev)
    nbytes = (batch + 7) // 8
    d_bm = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    d_all = torch.zeros(nbytes * world, dtype=torch.uint8, device=dev) if world > 1 else None
    torch.cuda.synchronize()

    # `--inflight M`: M contexts (own streams + workspaces, shared generator tables), one host thread
    # each; consecutive steps go to alternating contexts so the latency-bound tail of one batch (the
    # ~255-doubling Horner chain of its proof points) overlaps the chip-filling kernels of the next.
    # Every step is still one complete, independent verify call; K steps are timed as a whole.
    from concurrent.futures import ThreadPoolExecutor
    ctxs = [ctx] + [Context(local) for _ in range(max(1, args.inflight) - 1)]
    lanes = [ThreadPoolExecutor(max_workers=1) for _ in ctxs]

    def verify_on(c):
        return c.verify_batch_ps_dev(w["ps"], batch, d_dyn_sc, d_dyn_pt, d_dyn_off, batch * N_DYN,
                                     d_st_sc, None, d_st_off, batch * N_STATIC)

    def run_steps(n):
        futs = [lanes[i % len(ctxs)].submit(verify_on, ctxs[i % len(ctxs)]) for i in range(n)]
        bm = None
        for f in futs:
            bm = f.result()
            if world > 1:
                d_bm.copy_(torch.frombuffer(bytearray(bm), dtype=torch.uint8))
                dist.all_gather_into_tensor(d_all, d_bm)    # RCCL over xGMI: the per-shard accept bitmaps
        return bm

    bm = run_steps(max(args.warmup, len(ctxs)))
    for c in ctxs:
        c.profile_reset()
        c.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bm = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for c in ctxs:
        c.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = bytes(d_all.cpu().numpy().tobytes())
        assert gathered[rank * nbytes:(rank + 1) * nbytes] == bm

    assert bm == bitmap_of(w["expected"]), "accept bitmap differs from the constructed expectation"

    if rank == 0:
        prof = {}
        for c in ctxs:
            for k, v in c.profile_read().items():
                a = prof.get(k, (0, 0.0))
                prof[k] = (a[0] + v[0], a[1] + v[1])
        kern_ms = {k: v[1] / v[0] for k, v in prof.items() if v[0]}
        total_kernel_ms = sum(v[1] for v in prof.values()) / max(args.steps, 1)
        dom = max(kern_ms, key=kern_ms.get)
        dom_ms = kern_ms[dom]
        wbits = ctx.last_window_bits()                 # Pippenger width of the proof-point pipeline
        tbits = args.table_bits
        # Per-launch algorithmic bytes (64 B per proof-specific term, 32 B per
        # generator scalar, generator points amortised) and mixed additions of the kernels that carry them.
        per_kernel = {
            "k_static_accumulate": {"bytes": 32 * N_STATIC * batch,
                                    "madds": batch * N_STATIC * (255 // tbits + 1) if tbits else 0},
            "k_bucket_accumulate": {"bytes": (64 * N_DYN + (0 if tbits else 32 * N_STATIC)) * batch,
                                    "madds": batch * (N_DYN + (0 if tbits else N_STATIC)) * (255 // wbits + 1)},
        }
        info = per_kernel.get(dom, {"bytes": (64 * N_DYN + 32 * N_STATIC) * batch, "madds": 0})
        alg_bytes = info["bytes"]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        mads = info["madds"] * MADS_PER_MADD
        # solo pass: the same kernel with nothing else in flight (one context, profiling on)
        ctx.profile_reset()
        ctx.profile(True)
        for _ in range(5):
            verify_on(ctx)
        ctx.profile(False)
        solo = {k: v[1] / v[0] for k, v in ctx.profile_read().items() if v[0]}
        solo_ms = solo.get(dom, dom_ms)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom)
            except Exception:
                traffic = None
        line = {
            "metric": "ZkVM tx verifications/sec (batch)",
            "value": round(batch * world * args.steps / elapsed, 1),
            "unit": "tx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limb pairs of radix-2^51 (v_mad_u64_u32)",
            "data": "synthetic",
            "config": {"workload": "batch of %d 2-in/2-out-cloak-shaped R1CS verification MSMs per GPU "
                                   "(n=256, k=8, m=8: 549 terms, 514 on shared generators), ~1.5%% corrupted" % batch,
                       "tx_per_gpu": batch, "terms_per_tx": N_DYN + N_STATIC, "window_bits": wbits,
                       "generator_table_bits": args.table_bits, "calls_in_flight": len(ctxs),
                       "parallelism": "tx-sharded x%d, RCCL all-gather of accept bitmaps" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                         "avg_launch_ms_solo": round(solo_ms, 4),
                         "achieved_solo": round(alg_bytes / (solo_ms * 1e-3) / 1e9, 3),
                         "note": "avg_launch_ms is measured with %d verify calls in flight (kernels of different "
                                 "calls share the chip); *_solo is the same kernel alone" % len(ctxs),
                         "binding_resource": "integer VALU (v_mad_u64_u32) and random 128-B table gathers, "
                                             "not streaming HBM bandwidth",
                         "valu_int": {"achieved_Gmad_s": round(mads / (solo_ms * 1e-3) / 1e9, 1),
                                      "peak_Gmad_s": MAD_PEAK_GOPS,
                                      "frac": round(mads / (solo_ms * 1e-3) / 1e9 / MAD_PEAK_GOPS, 4),
                                      "basis": "solo launch"}},
            "kernel_ms_solo": {k: round(v, 4) for k, v in sorted(solo.items())},
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kern_ms.items())},
            "kernel_ms_total_per_step": round(total_kernel_ms, 4),
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(w, bm, batch)
        if world == 1 and not args.no_msm:
            line["msm_2p20"] = msm_microbench(ctx, torch, dev)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for ex in lanes:
        ex.shutdown()
    for c in ctxs[1:]:
        c.close()
    w["ps"].close()
    ctx.close()


if __name__ == "__main__":
    main()
