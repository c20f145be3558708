"""Soak of the ticket paths (round 5: host-memory tickets travel to an HBM twin of their staging area as they are staged; areas stay
taken until their batch is collected): many tickets of drawn sizes, host-memory and device tickets mixed in one queue, waited for
in a drawn order with a bounded number in flight, every verdict against the constructed expectation (which the GPU tests hold
against the oracle).  usage: ticket_soak.py [tickets=400] [seed=1] [merge=4096] [lanes=4]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import benched_randomness, benched_step, bits
from zkvm_amd import Context
from zkvm_amd.verifier import BlockVerifier, BulletproofGens
n_tickets = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
merge = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = random.Random(seed)
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=12)
bv = BlockVerifier(ctx, gens, batches_in_flight=lanes)
bv.set_merge(merge)
base = []
for s in range(6):
    txs, expected = benched_step(1024, 0, 64, 300 + s)
    base.append((b"".join(t[2] for t in txs), b"".join(t[3] for t in txs), benched_randomness(0, 300 + s, 1024), expected, len(txs[0][3])))
dev_keep, q, done = [], [], 0
t0 = time.perf_counter()
for k in range(n_tickets):
    com, proofs, r, exp, plen = base[rng.randrange(len(base))]
    n = rng.choice([1, 7, 64, 300, 1000, 1024, 1024, 1024])
    lo = rng.randrange(0, 1024 - n + 1)
    c, p, rr, want = com[256 * lo: 256 * (lo + n)], proofs[plen * lo: plen * (lo + n)], r[64 * lo: 64 * (lo + n)], exp[lo: lo + n]
    if rng.random() < 0.3:
        d = [ctx.to_device(x) for x in (c, p, rr)]
        dev_keep.append(d)
        q.append((bv.submit_dev(2, 2, n, d[0], d[1], plen, d[2]), want, n))
    else:
        q.append((bv.submit(2, 2, n, c, p, plen, rr), want, n))
    while len(q) > rng.choice([0, 3, 9, 17, 30]):
        tk, want, n = q.pop(rng.randrange(len(q)))
        assert bits(bv.wait(tk), n) == want, "a verdict differs (ticket %d)" % tk
        done += n
while q:
    tk, want, n = q.pop(rng.randrange(len(q)))
    assert bits(bv.wait(tk), n) == want, "a verdict differs (ticket %d)" % tk
    done += n
dt = time.perf_counter() - t0
print("ticket soak ok: %d tickets, %d transactions, seed %d, merge %d, %d lanes: %.2f s (%.0f tx/s, verdicts checked)" % (n_tickets, done, seed, merge, bv.lanes(), dt, done / dt))
bv.close()
for d in dev_keep:
    for x in d:
        ctx.free_device(x)
gens.close(); ctx.close()
