"""Prover calls under a grid of settings, one fresh process per setting (the library reads ZKGPU_PROVER_SLICES /
ZKGPU_STATIC_PIPE once): best-of-4 call time of zkgpu_cloak_prove_batch (2-in/2-out) and zkgpu_r1cs_prove_batch (the
1032-constraint program) -> one line per setting.   python3 tools/prover_sweep.py [child <kind> <batch> <bits>]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import hashlib, json, os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(kind, batch, bits):
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens
    ctx = Context(0)
    if kind == "cloak":
        from zkvm_amd.verifier import Prover
        gens = BulletproofGens(ctx, 256, table_bits=bits)
        rng = random.Random(1)
        qs, fs, seeds = [], [], []
        for i in range(batch):
            f = rng.randrange(2**250).to_bytes(32, "little")
            a, b = rng.randrange(2**40), rng.randrange(2**40)
            qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3]); fs.append([f] * 4); seeds.append(hashlib.sha256(b"p %d" % i).digest())
        pr = Prover(ctx, gens, host_threads=0)
        call = lambda: pr.prove(2, 2, qs, fs, seeds)
    else:
        from gpu_util import GADGET_LABEL, describe_ranges, gadget_witness
        from zkvm_amd.native import R1csDescription
        from zkvm_amd.verifier import R1csProver
        m, n1, n, labels, cons = describe_ranges(8)
        desc = R1csDescription(GADGET_LABEL, m, n1, n, labels, cons)
        gens = BulletproofGens(ctx, 512, table_bits=bits)
        rng = random.Random(3)
        vals, givens, seeds, mult_def = [], [], [], None
        for i in range(batch):
            values = [rng.randrange(1 << 64) for _ in range(8)]
            mult_def, given = gadget_witness(3, 8, values)
            vals.append(values); givens.append(given); seeds.append(hashlib.sha256(b"pp %d" % i).digest())
        pr = R1csProver(ctx, gens, desc, mult_def, host_threads=0)
        call = lambda: pr.prove(vals, givens, seeds)
    call(); call()
    ts = []
    for _ in range(4):
        call(); ts.append(pr.last_call_s)
    ctx.profile(True); ctx.profile_reset()
    call()
    prof = {k: round(v[1], 3) for k, v in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:8]}
    print(json.dumps({"kind": kind, "batch": batch, "table_bits": gens.points.table_bits(), "slices": os.environ.get("ZKGPU_PROVER_SLICES", "default"),
                      "lib": os.path.basename(os.path.dirname(os.environ.get("ZKGPU_LIB", "")) or "tree"),
                      "ms": round(min(ts) * 1e3, 2), "ms_all": [round(t * 1e3, 2) for t in ts], "proofs_per_s": round(batch / min(ts)),
                      "profiled_call_ms": round(pr.last_call_s * 1e3, 2), "kernel_ms": prof}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "slices"
    grid = []                                  # (kind, batch, table bits, slices, extra environment)
    if which == "slices":                      # round 5, first sweep (profiles/archive/r05b_*)
        for kind, batches in (("cloak", (2048, 4096)), ("program", (1024, 2048))):
            for batch in batches:
                for slices in (1, 2, 3, 4):
                    grid.append((kind, batch, 16, slices, {}))
        for kind, batch in (("cloak", 2048), ("program", 1024)):
            for bits in (12, 13, 14, 15):
                grid.append((kind, batch, bits, 2, {}))
    elif which == "lds":                       # fewer workgroups of k_static_accumulate per CU, so that phases fit beside it
        for kind, batch in (("cloak", 4096), ("program", 2048)):
            for slices in (0, 3, 4, 6):
                for kb in (0, 40, 64):
                    grid.append((kind, batch, 16, slices, {"ZKGPU_STATIC_LDS_KB": str(kb)}))
    elif which == "stages":                    # after the phases became stages (round 5, last): the slice counts again
        for kind, batch, sl in (("cloak", 8192, (1, 2, 3, 4, 6)), ("cloak", 4096, (1, 2, 4)), ("cloak", 2048, (1, 2, 3)),
                                ("program", 4096, (1, 2, 3, 4, 6)), ("program", 2048, (1, 2, 4)), ("program", 1024, (1, 2))):
            for slices in sl:
                grid.append((kind, batch, 16, slices, {}))
    elif which == "big":                       # larger calls, more slices
        for kind, batch, sl in (("cloak", 8192, (4, 6, 8)), ("cloak", 6144, (4, 6)), ("program", 4096, (4, 6, 8)), ("program", 3072, (3, 6))):
            for slices in sl:
                grid.append((kind, batch, 16, slices, {}))
    for kind, batch, bits, slices, extra in grid:
        env = dict(os.environ, **extra)
        if slices:
            env["ZKGPU_PROVER_SLICES"] = str(slices)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", kind, str(batch), str(bits)], env=env, capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        if line.startswith("{"):
            d = json.loads(line); d["env"] = extra
            print(json.dumps(d), flush=True)
        else:
            print("FAILED %s: %s" % ((kind, batch, bits, slices, extra), r.stderr[-400:]), flush=True)
