"""One zkgpu_tx_verify_batch call of 8192 DISTINCT transactions (gpu_util.built_transactions, 1 in 64 damaged), six times, with
the library's per-kernel HIP-event profile of the last call: what tools/profile_bench.sh runs under rocprofv3 for the
serialized-transaction path (kernel trace + stats, and one SQ_INSTS_VALU pass)."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpu_util import built_transactions
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
txs, exp = built_transactions(n, call=1, bad_every=64)
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=-1)
bv = BlockVerifier(ctx, gens)
bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
if os.environ.get("TXCHUNK"):
    bv.set_tx_chunk(int(os.environ["TXCHUNK"]))
blob, lens = b"".join(txs), np.asarray([len(t) for t in txs], dtype=np.uint64)
for _ in range(bv.lanes()):
    bv.verify_txs_packed(blob, lens)
for k in range(6):
    t0 = time.perf_counter()
    bm, st = bv.verify_txs_packed(blob, lens)
    dt = time.perf_counter() - t0
    print("call %d: %.2f ms, %.0f tx/s" % (k, dt * 1e3, n / dt))
assert [(bm[i // 8] >> (i % 8)) & 1 for i in range(n)] == exp
