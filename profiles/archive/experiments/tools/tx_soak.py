"""Soak of zkgpu_tx_verify_batch: random call sizes and chunk lengths over the committed (valid) transactions, a few of them
damaged per call (flipped signature bit: must be rejected, everybody else accepted).  usage: tx_soak.py [iterations]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, random, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import load_tx_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
base = load_tx_fixture()
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=12)
bv = BlockVerifier(ctx, gens)
bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
rng = random.Random(2026)
for it in range(iters):
    n = rng.choice([1, 7, 8, 9, 63, 64, 65, 1000, 1024, 3000, 8192, 12000, 20000])
    chunk = rng.choice([0, 0, 0, 97, 512, 1000, 2304, 4096, 9000])
    kept = rng.choice([None, None, 8, 1024, 1 << 17])
    txs = [base[(i * 7 + it) % 1024] for i in range(n)]
    bad = set(rng.sample(range(n), min(n, rng.choice([0, 1, 5]))))
    for i in bad:
        t = bytearray(txs[i])
        plen = struct.unpack("<I", t[24:28])[0]
        t[28 + plen + 32 + rng.randrange(31)] ^= 1 << rng.randrange(8)        # signature scalar s
        txs[i] = bytes(t)
    bv.set_tx_chunk(chunk)
    if kept is not None:
        bv.set_tx_statements_kept(kept)
    blob, lens = b"".join(txs), np.asarray([len(t) for t in txs], dtype=np.uint64)
    bm, st = bv.verify_txs_packed(blob, lens, rng.choice([0, 1, 3, 16]))
    got = [(bm[i // 8] >> (i % 8)) & 1 for i in range(n)]
    want = [0 if i in bad else 1 for i in range(n)]
    assert got == want, (it, n, chunk, kept, [i for i in range(n) if got[i] != want[i]][:10])
    assert all((st[i] == 0) == (want[i] == 1) for i in range(n)), (it, n, chunk)
    print("iteration %d: %d transactions, chunk %d, kept %s, %d damaged: ok" % (it, n, chunk, kept, len(bad)), flush=True)
print("soak ok")
