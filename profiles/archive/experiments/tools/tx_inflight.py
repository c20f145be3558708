"""Calls in flight on one verifier (zkgpu_tx_verify_submit / _wait): throughput by calls in flight and transactions per call.
usage: tx_inflight.py [per_call=1024] [in_flight=8] [calls=64] [lanes=0: the default]      (ZKGPU_TX_ROUNDS=1: one round at a time)"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import built_transactions
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, BlockVerifier
per_call = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 64
txs, exp = built_transactions(8192, call=1, bad_every=64)
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=14)
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bv = BlockVerifier(ctx, gens, batches_in_flight=lanes) if lanes else BlockVerifier(ctx, gens)
bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
for _ in range(bv.lanes()):
    bv.verify_txs(txs[:4096])
parts = 8192 // per_call
calls = []
for k in range(parts):
    part = txs[per_call * k: per_call * (k + 1)]
    calls.append((b"".join(part), np.asarray([len(x) for x in part], dtype=np.uint64)))
for rep in range(3):
    r0 = bv.tx_stats()
    q = collections.deque()
    t0 = time.perf_counter()
    for k in range(n_calls):
        if len(q) >= depth:
            bv.wait_txs(q.popleft())
        q.append(bv.submit_txs_packed(*calls[k % parts]))
    while q:
        bv.wait_txs(q.popleft())
    dt = time.perf_counter() - t0
    r1 = bv.tx_stats()
    print("%d per call, %d in flight, %d calls: %.2f ms, %.0f tx/s; %d rounds, %.2f calls per round" % (
        per_call, depth, n_calls, dt * 1e3, per_call * n_calls / dt, r1[0] - r0[0], (r1[1] - r0[1]) / max(1, r1[0] - r0[0])))
