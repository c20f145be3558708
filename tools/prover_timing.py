"""GPU: where the batch prover spends its time (ZKGPU_PROVER_TIMING=1 prints per phase)."""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ZKGPU_PROVER_TIMING"] = "1"
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Prover
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=13)
rng = random.Random(1)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
qs, fs, seeds = [], [], []
for i in range(batch):
    f = rng.randrange(2**250).to_bytes(32, "little")
    a, b = rng.randrange(2**40), rng.randrange(2**40)
    qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3]); fs.append([f] * 4)
    seeds.append(hashlib.sha256(b"t %d" % i).digest())
pr = Prover(ctx, gens, host_threads=int(sys.argv[2]) if len(sys.argv) > 2 else 16)
pr.prove(2, 2, qs[:4], fs[:4], seeds[:4])
t0 = time.perf_counter()
pr.prove(2, 2, qs, fs, seeds)
print("proofs/s", batch / (time.perf_counter() - t0))
