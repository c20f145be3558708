"""Two zkgpu_tx_verify_batch calls in flight: two host threads, each on a context and a verifier of its own (the host
stages of one call beside the device stages of the other).  usage: tx_concurrent.py [copies of the 1024 fixture per call] [lanes]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import load_tx_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
txs = load_tx_fixture() * rep
import numpy as np
blob, lens = b"".join(txs), np.asarray([len(t) for t in txs], dtype=np.uint64)      # (a Python list of lengths costs more than the call)
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
ctxs = [ctx, ctx.fork()]
bvs = [BlockVerifier(c, gens, batches_in_flight=lanes) for c in ctxs]
for bv in bvs:
    bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
    bv.verify_txs_packed(blob, lens)
calls = 6
def one(bv, out):
    for _ in range(calls):
        bm, st = bv.verify_txs_packed(blob, lens)
        assert not any(st)
t0 = time.perf_counter(); one(bvs[0], None); dt1 = time.perf_counter() - t0
th = [threading.Thread(target=one, args=(bv, None)) for bv in bvs]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt2 = time.perf_counter() - t0
print("one call at a time: %.0f tx/s; two in flight: %.0f tx/s (%d transactions per call, %d lanes per verifier)"
      % (calls * len(txs) / dt1, 2 * calls * len(txs) / dt2, len(txs), lanes))
