"""Per-call times of zkgpu_tx_verify_batch on one verifier.  usage: tx_calls.py [copies] [lanes] [second verifier: 0/1]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import load_tx_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
second = int(sys.argv[3]) if len(sys.argv) > 3 else 0
txs = load_tx_fixture() * rep
blob, lens = b"".join(txs), np.asarray([len(t) for t in txs], dtype=np.uint64)
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
bv = BlockVerifier(ctx, gens, batches_in_flight=lanes)
other = BlockVerifier(ctx.fork(), gens, batches_in_flight=lanes) if second else None
bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); bm, st = bv.verify_txs_packed(blob, lens); ts.append((time.perf_counter() - t0) * 1e3)
    assert not any(st)
    time.sleep(0.02)                                     # (a gap between the calls: a kernel trace then shows them apart)
print("lanes kept %s, second verifier %d: per call ms %s" % (bv.queue_info(), second, " ".join("%.1f" % t for t in ts)))
