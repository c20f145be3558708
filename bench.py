#!/usr/bin/env python3
"""bench.py -- batch verification throughput of ZkVM cloak transactions on MI355X.

One "step" = one complete `r1cs::Verifier::verify` of every transaction of one batch, on the
device, from inputs resident in HBM (zkgpu_cloak_verify_batch_gpu_dev): Merlin transcript
replay, verification scalars (inner-product-argument s vector, constraint flattening, g_i /
h_i), decompression of the proof points, the 549-term multiscalar multiplication per
transaction and the ristretto identity test -> accept bitmap (copied to the host).
Workload = BASELINE.json configs[1]: 1024 2-in/2-out cloak transactions per GPU.  The proofs
are REAL Bulletproofs R1CS proofs of the cloak statement (tests/golden/cloak_2x2_proofs.bin:
64 proofs from the oracle prover, committed as data); transaction i verifies proof
i mod 64 under its own verifier randomness r_i, so all 1024 verification equations differ
(n = 256 multipliers, k = 8, m = 8 commitments: 35 proof-specific points + 514 shared
generators per transaction).  ~1.5 % of the transactions are corrupted (undecodable
commitment, wrong IPA scalar, someone else's proof) so the accept bitmap is not trivial.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
One process per GPU; shards are independent (weak scaling: 1024 tx per GPU); the only
collective is the RCCL all-gather of the per-shard accept bitmaps.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel vs the HBM
roofline the north-star names, plus the integer-ALU figure that actually binds),
"cpu_baseline" (the CPU oracle's full verifier -- a port of the reference's algorithm; the
reference itself is not mounted -- on this box's host cores), "msm_boundary" (the
multiscalar-multiplication tail alone), "host_memory" (the same calls fed from host memory),
"msm_2p20" (BASELINE configs[2] microbench).
"""
from __future__ import annotations

import os

# Each batch in flight has a stream of its own for its latency-bound kernels; the HIP runtime maps
# streams onto 4 hardware queues by default, which would serialise those streams again.  Must be set
# before the runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import argparse
import hashlib
import json
import os
import struct
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
MAD_PEAK_GOPS = 33864.9                      # measured v_mad_u64_u32 rate, profiles/r01_valu_rates.txt
MADS_PER_MADD = 700                          # 7 field multiplications x 100 (field.hpp)
BAD_POINT = bytes.fromhex("01" + "00" * 31)


def shake(tag: bytes, n: int) -> bytes:
    return hashlib.shake_256(SEED.to_bytes(4, "little") + tag).digest(n)


def load_fixture():
    path = os.path.join(ROOT, "tests", "golden", "cloak_2x2_proofs.bin")
    raw = open(path, "rb").read()
    assert raw[:8] == b"ZKCLOAK1"
    count, n_in, n_out, plen = struct.unpack("<IIII", raw[8:24])
    w = 64 * (n_in + n_out)
    rec = w + plen
    return [(raw[24 + rec * i: 24 + rec * i + w], raw[24 + rec * i + w: 24 + rec * (i + 1)]) for i in range(count)], n_in, n_out


def build_workload(ctx, batch: int, rank: int, table_bits: int, host_threads: int):
    """Real proofs -> verification equations, with the product's own host verifier."""
    from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
    fixture, n_in, n_out = load_fixture()
    gens = BulletproofGens(ctx, N_MULT, table_bits=table_bits)
    txs, expected = [], []
    for i in range(batch):
        com, proof = fixture[(i + 7 * rank) % len(fixture)]
        ok = 1
        if i % 64 == 7:
            ok = 0
            c = (i // 64) % 3
            if c == 0:      # commitment that is not a ristretto255 encoding
                com = com[:96] + BAD_POINT + com[128:]
            elif c == 1:    # IPA scalar a off by one (still canonical)
                a = (int.from_bytes(proof[-64:-32], "little") + 1) % L
                proof = proof[:-64] + a.to_bytes(32, "little") + proof[-32:]
            else:           # a valid proof of a different statement
                proof = fixture[(i + 7 * rank + 1) % len(fixture)][1]
        txs.append(CloakTx(n_in, n_out, com, proof))
        expected.append(ok)
    r_bytes = shake(b"verifier-r|%d" % rank, 64 * batch)
    v = Verifier(ctx, gens, host_threads=host_threads)
    t0 = time.perf_counter()
    prep = v.prepare(txs, r_bytes)
    prep_s = time.perf_counter() - t0
    assert all(prep["wellformed"]), "bench corruptions keep proofs well-formed so the GPU sees every row"
    assert prep["dyn_off"][-1] == batch * N_DYN and prep["st_off"][-1] == batch * N_STATIC
    prep.update({"gens": gens, "txs": txs, "r_bytes": r_bytes, "expected": expected, "verifier": v,
                 "prepare_s": prep_s})
    return prep


def bitmap_of(bits) -> bytes:
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i // 8] |= 1 << (i % 8)
    return bytes(out)


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


def cpu_baseline(ctx, w, gpu_bitmap: bytes, batch: int):
    """Time the CPU oracle's full verifier (kind "port": transcript replay + scalars + MSM with the same
    radix-2^51 field and Straus/Pippenger split as the reference's dalek back end; the Rust reference
    itself is not mounted) on the same proof bytes, and compare every accept bit with the GPU's."""
    from oracle import binding as oracle
    cores = usable_cores(oracle.max_threads())
    tx0 = w["txs"][0]
    plen = len(tx0.proof)
    com = b"".join(t.commitments for t in w["txs"])
    proofs = b"".join(t.proof for t in w["txs"])
    wcom = 64 * (tx0.n_in + tx0.n_out)
    one = 32
    t0 = time.perf_counter()
    acc1 = oracle.cloak_verify_batch(com[: wcom * one], tx0.n_in, tx0.n_out, proofs[: plen * one], plen,
                                     w["r_bytes"][: 64 * one], threads=1)
    t1 = time.perf_counter() - t0
    reps = 0
    t0 = time.perf_counter()
    while True:
        acc = oracle.cloak_verify_batch(com, tx0.n_in, tx0.n_out, proofs, plen, w["r_bytes"], threads=cores)
        reps += 1
        if time.perf_counter() - t0 > 8.0 or reps >= 50:
            break
    tall = time.perf_counter() - t0
    gpu_bits = [(gpu_bitmap[i // 8] >> (i % 8)) & 1 for i in range(batch)]
    assert list(acc) == gpu_bits, "GPU accept bitmap differs from the CPU oracle"
    assert list(acc1) == gpu_bits[:one]
    # the oracle's PROVER on the same cores (baseline of the `prover` leg)
    n_pr = 4 * cores
    t0 = time.perf_counter()
    oracle.cloak_prove_batch(n_pr, 2, 2, b"bench prover cpu".ljust(32, b"\0"), threads=cores)
    prover_rate = n_pr / (time.perf_counter() - t0)
    return {"value": round(batch * reps / tall, 1), "unit": "tx/s", "cores": cores, "kind": "port",
            "value_1core": round(one / t1, 2), "prover_proofs_per_s": round(prover_rate, 1),
            "sample": "%d x the full %d-tx batch (oracle Verifier: transcript replay + scalars + 549-term MSM per tx, "
                      "same proof bytes and verifier randomness as the GPU step) on %d OpenMP threads = this box's "
                      "cgroup CPU quota (%.1f s); 1-core figure on %d tx (%.1f s); all %d accept bits compared "
                      "with the GPU's" % (reps, batch, cores, tall, one, t1, batch)}


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


def prover_microbench(ctx, w, host_threads: int, batch: int = 512):
    """BASELINE configs[4] (prover side): `batch` 2-in/2-out cloak proofs, host threads for the transcripts and
    the witness / polynomial algebra, every multiscalar multiplication on the generator tables."""
    import random
    from zkvm_amd.verifier import Prover, Verifier
    rng = random.Random(SEED)
    qs, fs, seeds = [], [], []
    for i in range(batch):
        f = rng.randrange(2**250).to_bytes(32, "little")
        a, b = rng.randrange(2**40), rng.randrange(2**40)
        qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3])
        fs.append([f] * 4)
        seeds.append(hashlib.sha256(b"bench prover %d" % i).digest())
    pr = Prover(ctx, w["gens"], host_threads=host_threads)
    pr.prove(2, 2, qs[:8], fs[:8], seeds[:8])
    t0 = time.perf_counter()
    txs = pr.prove(2, 2, qs, fs, seeds)
    dt = time.perf_counter() - t0
    v = Verifier(ctx, w["gens"])
    bm = v.verify_bitmap_gpu(txs, shake(b"prover-r", 64 * batch))
    v.close()
    assert bm == bitmap_of([1] * batch), "a proof of the GPU prover did not verify"
    out = {"proofs_per_s": round(batch / dt, 1), "batch": batch, "ms_per_proof": round(dt / batch * 1e3, 4),
           "host_threads": host_threads, "msm_terms_per_proof": 2 * 4 * 2 + 3 * 273 + 3 * 29 + 5 * 2 + 16 * 513,
           "note": "zkgpu_cloak_prove_batch: provers in lockstep on host threads (host-bound: scalar algebra of the "
                   "coefficient-vector inner-product argument), all MSMs in 13 zkgpu_msm_ps_batch calls on the tables; "
                   "every proof verified by the device-side verifier"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="transactions per GPU")
    ap.add_argument("--table-bits", type=int, default=int(os.environ.get("ZKGPU_TABLE_BITS", "16")),
                    help="window width of the fixed-base generator tables (0 = no tables: Pippenger for every term)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ZKGPU_INFLIGHT", "6")),
                    help="independent verify calls in flight per GPU (contexts x host threads)")
    ap.add_argument("--group", type=int, default=int(os.environ.get("ZKGPU_GROUP", "16")),
                    help="transactions per group check (zkgpu_set_group_size); 1 = every transaction on its own")
    ap.add_argument("--lean", action="store_true",
                    help="the timed steps and the solo pass only (no extra legs): what tools/profile_bench.sh profiles")
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
    # ZKGPU_BENCH_SHARE_GPU=1 (testing the N > 1 code path on a 1-GPU box): every rank uses device 0 and the
    # bitmaps travel over gloo from host memory instead of RCCL (which refuses two ranks on one device)
    share_gpu = os.environ.get("ZKGPU_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = torch.device("cpu") if share_gpu else dev
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zkvm_amd import Context
    ctx = Context(local)
    batch = args.batch
    host_threads = max(1, usable_cores(os.cpu_count() or 1) // max(1, world))
    w = build_workload(ctx, batch, rank, args.table_bits, host_threads)
    ps = w["gens"].points

    def to_dev(b, dtype=torch.uint8):
        return torch.frombuffer(bytearray(b), dtype=dtype).to(dev)

    d_dyn_sc, d_dyn_pt, d_st_sc = to_dev(w["dyn_sc"]), to_dev(w["dyn_pt"]), to_dev(w["st_sc"])
    d_st_idx = torch.tensor(w["st_idx"], dtype=torch.int32, device=dev)
    d_dyn_off = torch.tensor(w["dyn_off"], dtype=torch.int64, device=dev)
    d_st_off = torch.tensor(w["st_off"], dtype=torch.int64, device=dev)
    nbytes = (batch + 7) // 8
    d_bm = torch.zeros(nbytes, dtype=torch.uint8, device=coll_dev)
    d_all = torch.zeros(nbytes * world, dtype=torch.uint8, device=coll_dev) if world > 1 else None
    torch.cuda.synchronize()

    # `--inflight M`: M batches in flight.  Each has its own forked context (workspace + a light stream
    # for its latency-bound kernels: the Merlin replay, the 255-doubling Horner tail); the chip-filling
    # kernels of all of them go first-in first-out through the parent's two streams (zkgpu_ctx_fork).
    # One host thread submits step i + M only after collecting step i.  Every step is still one
    # complete, independent verification of the whole batch; K steps are timed as a whole.
    ctx.set_group_size(args.group)               # forks inherit it
    lanes_env = min(max(int(os.environ.get("ZKGPU_LANES", "2")), 1), 4)
    max_inflight = 1 + min(9, 19 - 3 * lanes_env)          # the library's fork limit (hardware queues)
    ctxs = [ctx] + [ctx.fork() for _ in range(min(max(1, args.inflight), max_inflight) - 1)]

    # THE STEP: the complete r1cs::Verifier::verify of every transaction of the batch, on the device,
    # from commitments + proof bytes + verifier randomness resident in HBM: Merlin transcript replay
    # (k_transcript), verification scalars incl. the inner-product-argument s vector (k_prepare), point
    # decompression, the multiscalar multiplications and the identity test -> accept bitmap.
    from zkvm_amd.verifier import Verifier
    tx0 = w["txs"][0]
    proof_len = len(tx0.proof)
    d_com = to_dev(b"".join(t.commitments for t in w["txs"]))
    d_proofs = to_dev(b"".join(t.proof for t in w["txs"]))
    d_r = to_dev(w["r_bytes"])
    gv = Verifier(ctx, w["gens"])
    torch.cuda.synchronize()

    def submit_verify(c):
        gv.submit_packed_gpu_dev(tx0.n_in, tx0.n_out, batch, d_com, d_proofs, proof_len, d_r, ctx=c)

    def submit_msm_only(c):   # the MSM boundary alone: scalars prepared beforehand (by the host verifier)
        c.verify_batch_ps_submit_dev(ps, batch, d_dyn_sc, d_dyn_pt, d_dyn_off, batch * N_DYN,
                                     d_st_sc, d_st_idx, d_st_off, batch * N_STATIC)

    def collect(c, gather=True):
        bm = c.verify_wait()
        if world > 1 and gather:
            d_bm.copy_(torch.frombuffer(bytearray(bm), dtype=torch.uint8))
            dist.all_gather_into_tensor(d_all, d_bm)    # RCCL over xGMI: the per-shard accept bitmaps
        return bm

    host_time = {"submit": 0.0, "n": 0}

    def run_steps(n, submit=None, gather=True):
        # gather=False: the rank-0-only extra legs (no collective: the other ranks are not in them)
        submit = submit or submit_verify
        depth = len(ctxs)
        bm = None
        for i in range(n):
            c = ctxs[i % depth]
            if i >= depth:
                bm = collect(c, gather)
            ts = time.perf_counter()
            submit(c)
            host_time["submit"] += time.perf_counter() - ts
            host_time["n"] += 1
        for i in range(max(n - depth, 0), n):
            bm = collect(ctxs[i % depth], gather)
        return bm

    bm = run_steps(max(args.warmup, len(ctxs)))
    # HIP events around every launch of ONE of the contexts in flight (every len(ctxs)-th step): the
    # per-kernel durations of the roofline object are measured inside the timed region without
    # fencing every kernel of every batch
    prof_ctxs = ctxs[:1]
    for c in prof_ctxs:
        c.profile_reset()
        c.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_time.update(submit=0.0, n=0)
    bm = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    submit_ms = host_time["submit"] / max(host_time["n"], 1) * 1e3
    for c in prof_ctxs:
        c.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = bytes(d_all.cpu().numpy().tobytes())
        assert gathered[rank * nbytes:(rank + 1) * nbytes] == bm

    assert bm == bitmap_of(w["expected"]), "accept bitmap differs from the constructed expectation"

    if rank == 0:
        prof = {}
        for c in prof_ctxs:
            for k, v in c.profile_read().items():
                a = prof.get(k, (0, 0.0))
                prof[k] = (a[0] + v[0], a[1] + v[1])
        kern_ms = {k: v[1] / v[0] for k, v in prof.items() if v[0]}
        profiled_steps = max(1, (args.steps + len(ctxs) - 1) // len(ctxs))
        total_kernel_ms = sum(v[1] for v in prof.values()) / profiled_steps
        # Dominant kernel = the one that issues most of the step's VALU work, not the longest-lived one:
        # with several calls in flight the latency-bound tail kernels (k_msm_finish: 16-64 wavefronts
        # walking a 255-doubling chain) show long durations while occupying a sliver of the chip.
        # SQ_INSTS_VALU per launch (profiles/r01c_pmc_SQ_WAVE_CYCLES.txt): k_static_accumulate 203 M,
        # k_bucket_accumulate / k_small_msm_windows ~100 M, k_msm_finish 15 M.
        dom = "k_static_accumulate" if (args.table_bits and "k_static_accumulate" in kern_ms) else "k_bucket_accumulate"
        if dom not in kern_ms:
            dom = max(kern_ms, key=kern_ms.get)
        dom_ms = kern_ms[dom]
        wbits = ctx.last_window_bits()                 # Pippenger width of the proof-point pipeline
        tbits = args.table_bits
        # Per-launch algorithmic bytes (64 B per proof-specific term, 32 B per
        # generator scalar, generator points amortised) and mixed additions of the kernels that carry them.
        # k_static_accumulate runs twice per step when the batch is checked in groups: once over the
        # n_groups summed checks, once over the transactions of the groups that failed
        n_win = (255 // tbits + 1) if tbits else 0
        if args.group > 1:
            g = min(args.group, batch)
            n_groups = (batch + g - 1) // g
            recheck = sum(min(g, batch - G * g) for G in range(n_groups) if not all(w["expected"][G * g: (G + 1) * g]))
            sa_terms, sa_launches = (n_groups + recheck) * N_STATIC, 2
        else:
            n_groups, recheck = batch, 0
            sa_terms, sa_launches = batch * N_STATIC, 1
        per_kernel = {
            "k_static_accumulate": {"bytes": 32 * sa_terms / sa_launches, "madds": sa_terms * n_win / sa_launches},
            "k_bucket_accumulate": {"bytes": (64 * N_DYN + (0 if tbits else 32 * N_STATIC)) * batch,
                                    "madds": batch * (N_DYN + (0 if tbits else N_STATIC)) * (255 // wbits + 1)},
        }
        info = per_kernel.get(dom, {"bytes": (64 * N_DYN + 32 * N_STATIC) * batch, "madds": 0})
        alg_bytes = info["bytes"]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        mads = info["madds"] * MADS_PER_MADD
        # solo pass: every kernel of a batch alone on the chip (one context, one stream, profiling on)
        ctx.profile_reset()
        ctx.set_serial(True)
        ctx.profile(True)
        for _ in range(5):
            submit_verify(ctx)
            ctx.verify_wait()
        ctx.profile(False)
        ctx.set_serial(False)
        solo = {k: v[1] / v[0] for k, v in ctx.profile_read().items() if v[0]}
        solo_ms = solo.get(dom, dom_ms)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom)
            except Exception:
                traffic = None
        e2e_s = e2e_gpu_s = msm_only_s = per_tx_s = float("nan")
        if not args.lean:
            # proof bytes -> accept bits, host half included (Merlin replay etc. on the host cores)
            v = w["verifier"]
            t0 = time.perf_counter()
            bm_e2e = v.verify_bitmap(w["txs"], w["r_bytes"])
            e2e_s = time.perf_counter() - t0
            assert bm_e2e == bm
            # the same call with its inputs in HOST memory (PCIe copies + python marshalling included)
            packed_com = b"".join(t.commitments for t in w["txs"])
            packed_proofs = b"".join(t.proof for t in w["txs"])
            n_e2e = 6 * len(ctxs)
            def submit_host(c):
                gv.submit_packed_gpu(tx0.n_in, tx0.n_out, batch, packed_com, packed_proofs, proof_len, w["r_bytes"], ctx=c)
            assert run_steps(len(ctxs), submit_host, gather=False) == bm
            t0 = time.perf_counter()
            outs = [run_steps(n_e2e, submit_host, gather=False)]
            e2e_gpu_s = (time.perf_counter() - t0) / n_e2e
            assert all(o == bm for o in outs)
            # the multiscalar-multiplication boundary alone (scalars prepared beforehand by the host verifier)
            assert run_steps(len(ctxs), submit_msm_only, gather=False) == bm
            t0 = time.perf_counter()
            run_steps(args.steps, submit_msm_only, gather=False)
            msm_only_s = (time.perf_counter() - t0) / args.steps
            # the same complete verification with every transaction checked on its own (no group checks)
            for c in ctxs:
                c.set_group_size(1)
            assert run_steps(len(ctxs), gather=False) == bm
            t0 = time.perf_counter()
            run_steps(args.steps, gather=False)
            per_tx_s = (time.perf_counter() - t0) / args.steps
            for c in ctxs:
                c.set_group_size(args.group)
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
            "data": "synthetic: 64 real R1CS proofs of the 2-in/2-out cloak statement (committed fixture), "
                    "each verified under per-transaction verifier randomness; ~1.5% corrupted",
            "config": {"workload": "batch of %d 2-in/2-out cloak tx per GPU, complete r1cs::Verifier::verify on the device "
                                   "from commitments + R1CSProof bytes + verifier randomness resident in HBM: Merlin "
                                   "replay, verification scalars (IPA s vector, constraint flattening), decompression, "
                                   "the 549-term mega_check MSM (n=256, k=8, m=8; 514 terms on shared generators), "
                                   "identity test -> accept bitmap" % batch,
                       "tx_per_gpu": batch, "terms_per_tx": N_DYN + N_STATIC, "window_bits": wbits,
                       "generator_table_bits": tbits, "calls_in_flight": len(ctxs), "group_size": args.group,
                       "parallelism": "tx-sharded x%d, RCCL all-gather of accept bitmaps" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                         "avg_launch_ms_solo": round(solo_ms, 4),
                         "achieved_solo": round(alg_bytes / (solo_ms * 1e-3) / 1e9, 3),
                         "note": "avg_launch_ms is measured with %d verify calls in flight (kernels of different "
                                 "calls share the chip); *_solo is the same kernel alone" % len(ctxs),
                         "binding_resource": "random 96-B gathers from the generator tables (one per mixed addition, "
                                             "no reuse: %.1f GB of tables) and integer VALU (v_mad_u64_u32); "
                                             "not streaming HBM bandwidth" % (int(ctx.lib.zkgpu_pointset_table_bytes(w["gens"].points.h)) / 1e9),
                         "table_gather": {"bytes_per_launch": int(info["madds"] * 96),
                                          "GBs_solo": round(info["madds"] * 96 / (solo_ms * 1e-3) / 1e9, 1),
                                          "frac_of_hbm_peak_solo": round(info["madds"] * 96 / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "valu_int": {"achieved_Gmad_s": round(mads / (solo_ms * 1e-3) / 1e9, 1),
                                      "peak_Gmad_s": MAD_PEAK_GOPS,
                                      "frac": round(mads / (solo_ms * 1e-3) / 1e9 / MAD_PEAK_GOPS, 4),
                                      "basis": "solo launch"}},
            "kernel_ms_per_step": {k: round(x, 4) for k, x in sorted(kern_ms.items())},
            "kernel_ms_solo": {k: round(x, 4) for k, x in sorted(solo.items())},
            "kernel_ms_total_per_step": round(total_kernel_ms, 4),
            "host_submit_ms_per_step": round(submit_ms, 4),
            "per_tx_checks": {"tx_per_s": round(batch / per_tx_s, 1), "ms_per_step": round(per_tx_s * 1e3, 4),
                              "note": "the same step with zkgpu_set_group_size(1): 1024 independent 549-term multiscalar "
                                      "multiplications instead of %d group checks + individual re-checks of the groups "
                                      "that hold a bad transaction" % ((batch + max(args.group, 1) - 1) // max(args.group, 1))},
            "msm_boundary": {"tx_per_s": round(batch / msm_only_s, 1), "ms_per_step": round(msm_only_s * 1e3, 4),
                             "note": "zkgpu_verify_batch_ps_dev alone: decompress + MSM + identity test on scalars "
                                     "prepared beforehand (the argument list of dalek's mega_check resident in HBM)"},
            "host_memory": {"gpu_resident_tx_per_s": round(batch / e2e_gpu_s, 1),
                            "host_prepared_tx_per_s": round(batch / e2e_s, 1), "host_threads": host_threads,
                            "host_prepare_ms_per_batch": round(w["prepare_s"] * 1e3, 2),
                            "note": "proof bytes in host memory -> accept bits, PCIe copies and python marshalling "
                                    "included.  gpu_resident: zkgpu_cloak_verify_batch_gpu (everything after the copy on "
                                    "the device).  host_prepared: zkgpu_cloak_verify_batch (transcript replay and scalar "
                                    "preparation on %d host threads, host-bound).  Neither is `value`." % host_threads},
        }
        if args.lean:
            for key in ("per_tx_checks", "msm_boundary", "host_memory"):
                line.pop(key, None)
        if world == 1 and not args.no_cpu and not args.lean:
            line["cpu_baseline"] = cpu_baseline(ctx, w, bm, batch)
        if world == 1 and not args.no_msm and not args.lean:
            line["prover"] = prover_microbench(ctx, w, host_threads)
            if "cpu_baseline" in line:
                line["prover"]["cpu_oracle_proofs_per_s"] = line["cpu_baseline"].get("prover_proofs_per_s")
        if world == 1 and not args.no_msm and not args.lean:
            line["msm_2p20"] = msm_microbench(ctx, torch, dev)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gv.close()
    for c in ctxs[1:]:
        c.close()
    w["gens"].close()
    ctx.close()


if __name__ == "__main__":
    main()
