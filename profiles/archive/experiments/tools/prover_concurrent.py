"""Device prover with several calls in flight: N host threads, each proving batches on a context of its own
(forks of one context: shared tables).  usage: python tools/prover_concurrent.py [threads] [batch] [rounds]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import hashlib, os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Prover

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
rng = random.Random(1)
qs, fs, seeds = [], [], []
for i in range(batch):
    f = rng.randrange(2**250).to_bytes(32, "little")
    a, b = rng.randrange(2**40), rng.randrange(2**40)
    qs.append([a, b, (a + b) // 3, a + b - (a + b) // 3]); fs.append([f] * 4); seeds.append(hashlib.sha256(b"p %d" % i).digest())
ctxs = [ctx] + [ctx.fork() for _ in range(nthreads - 1)]
provers = [Prover(c, gens, host_threads=max(1, 16 // nthreads)) for c in ctxs]
import ctypes as C
qa = (C.c_uint64 * (4 * batch))(*[q for row in qs for q in row])
fl = b"".join(f for row in fs for f in row)
sd = b"".join(seeds)
for p in provers:
    p.prove_packed(2, 2, batch, qa, fl, sd)
    p.prove_packed(2, 2, batch, qa, fl, sd)
spent = [0.0] * nthreads
def work(k):
    for _ in range(rounds):
        provers[k].prove_packed(2, 2, batch, qa, fl, sd)
        spent[k] += provers[k].last_call_s
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print("%d threads x %d rounds x %d proofs: %.1f ms wall, %.0f proofs/s (library calls: %.1f ms per call on average; contiguous inputs, no per-proof Python work)"
      % (nthreads, rounds, batch, dt * 1e3, nthreads * rounds * batch / dt, sum(spent) / (nthreads * rounds) * 1e3))
