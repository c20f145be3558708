"""zkgpu_tx_verify_batch on the committed 1024 transactions (per-stage times with ZKGPU_PROVER_TIMING=1).
usage: tx_bench.py [copies of the fixture per call] [block chunk] [tx chunk]"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import load_tx_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
txs = load_tx_fixture() * rep
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=16)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bv = BlockVerifier(ctx, gens, chunk=chunk)
bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
if len(sys.argv) > 3:
    bv.set_tx_chunk(int(sys.argv[3]))
HT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if os.environ.get("TX_BENCH_TRANSCRIPT_MODE"):        # experiment: the lanes' transcript replay (0 automatic, 1 lane, 2 cooperative)
    for i in range(bv.lanes()):
        bv.lane(i).set_transcript_mode(int(os.environ["TX_BENCH_TRANSCRIPT_MODE"]))
bv.verify_txs(txs[:64])
blob, lens = b"".join(txs), [len(t) for t in txs]
for _ in range(4):
    t0 = time.perf_counter()
    bm, st = bv.verify_txs_packed(blob, lens)
    dt = time.perf_counter() - t0
    print("%.2f ms, %.0f tx/s (%d transactions per call)" % (dt * 1e3, len(txs) / dt, len(txs)), file=sys.stderr)
assert os.environ.get("ZKGPU_TEST_TX_FREE_HASHING") == "1" or not any(st)      # (a -DZK_MEASURE_FREE_HASHING build with that variable set makes every signature fail)
# the library call alone (offsets and output buffers made beforehand)
offs = np.zeros(len(lens) + 1, dtype=np.uint64)
np.cumsum(np.asarray(lens, dtype=np.uint64), out=offs[1:])
bmb, stb = C.create_string_buffer((len(lens) + 7) // 8), C.create_string_buffer(len(lens))
for _ in range(3):
    t0 = time.perf_counter()
    rc = bv.lib.zkgpu_tx_verify_batch(bv.h, len(lens), blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), HT, bmb, stb)
    dt = time.perf_counter() - t0
    print("library call alone: %.2f ms, %.0f tx/s (rc %d)" % (dt * 1e3, len(txs) / dt, rc), file=sys.stderr)
ctx.profile(True); ctx.profile_reset()
bm, st = bv.verify_txs(txs)
ctx.profile(False)
for name, (n, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1])[:14]:
    print('%-24s %4d launches %9.3f ms' % (name, n, ms), file=sys.stderr)
