"""Per-kernel time of one zkgpu_r1cs_prove_batch call on the 1032-constraint program (8 x 64-bit range proofs)."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1"); _os.environ.setdefault("ZKGPU_PROVER_SLICES", "1")   # (the counter tables describe UNSLICED launches)   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import hashlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import GADGET_LABEL, describe_ranges, gadget_witness
from zkvm_amd import Context
from zkvm_amd.native import R1csDescription
from zkvm_amd.verifier import BulletproofGens, R1csProver
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = Context(0)
m, n1, n, labels, cons = describe_ranges(8)
desc = R1csDescription(GADGET_LABEL, m, n1, n, labels, cons)
gens = BulletproofGens(ctx, 512, table_bits=16)
rng = random.Random(3)
vals, givens, seeds, mult_def = [], [], [], None
for i in range(batch):
    values = [rng.randrange(1 << 64) for _ in range(8)]
    mult_def, given = gadget_witness(3, 8, values)
    vals.append(values); givens.append(given); seeds.append(hashlib.sha256(b"pp %d" % i).digest())
pr = R1csProver(ctx, gens, desc, mult_def, host_threads=0)
pr.prove(vals[:8], givens[:8], seeds[:8])
pr.prove(vals, givens, seeds)
print("call %.1f ms for %d proofs (%.0f proofs/s)" % (pr.last_call_s * 1e3, batch, batch / pr.last_call_s))
ctx.profile(True); ctx.profile_reset()
pr.prove(vals, givens, seeds)
tot = 0.0
for name, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
    print("%-24s %4d launches %9.3f ms" % (name, cnt, ms)); tot += ms
print("sum of kernels %.1f ms" % tot)
print("PMC_META full_calls=2 batch=%d (plus one call of 8 statements: 0.4 %% of a full call)" % batch)
