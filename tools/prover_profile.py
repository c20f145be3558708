"""Per-kernel time of one zkgpu_cloak_prove_batch call (device prover), from the library's profile hooks."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1"); _os.environ.setdefault("ZKGPU_PROVER_SLICES", "1")   # (the counter tables describe UNSLICED launches)   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import hashlib, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Prover

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
rng = random.Random(1)
qs, fs, seeds = [], [], []
for i in range(batch):
    f = rng.randrange(2**250).to_bytes(32, "little")
    a, b = rng.randrange(2**40), rng.randrange(2**40)
    qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3]); fs.append([f] * 4); seeds.append(hashlib.sha256(b"p %d" % i).digest())
pr = Prover(ctx, gens, host_threads=0)
pr.prove(2, 2, qs[:8], fs[:8], seeds[:8])
pr.prove(2, 2, qs, fs, seeds)
print("call %.1f ms for %d proofs (%.0f proofs/s)" % (pr.last_call_s * 1e3, batch, batch / pr.last_call_s))
ctx.profile(True); ctx.profile_reset()
pr.prove(2, 2, qs, fs, seeds)
print("profiled call %.1f ms" % (pr.last_call_s * 1e3))
tot = 0.0
for name, (n, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
    print("%-24s %4d launches %9.3f ms" % (name, n, ms)); tot += ms
print("sum of kernels %.1f ms" % tot)
print("PMC_META full_calls=2 batch=%d (plus one call of 8 statements: 0.4 %% of a full call)" % batch)
