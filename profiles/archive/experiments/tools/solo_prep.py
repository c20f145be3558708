import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import sys, os, time, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
ctx = Context(0)
fixture, n_in, n_out = bench.load_fixture()
B = 1024
gens = BulletproofGens(ctx, 256, table_bits=13)
txs = [CloakTx(n_in, n_out, *fixture[i % 64]) for i in range(B)]
com = b"".join(t.commitments for t in txs); proofs = b"".join(t.proof for t in txs)
r = hashlib.shake_256(b"r").digest(64 * B)
v = Verifier(ctx, gens)
bm = v.verify_packed_gpu(n_in, n_out, B, com, proofs, 1025, r)
print("accepts", sum(bin(x).count("1") for x in bm))
ctx.profile_reset(); ctx.profile(True)
t0 = time.perf_counter()
for _ in range(10): v.verify_packed_gpu(n_in, n_out, B, com, proofs, 1025, r)
dt = (time.perf_counter() - t0) / 10
ctx.profile(False)
print("solo ms per batch", round(dt * 1e3, 3))
for k, (n, ms) in sorted(ctx.profile_read().items(), key=lambda x: -x[1][1]):
    print("%-24s %8.4f ms" % (k, ms / n))
