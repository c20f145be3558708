"""Generates tests/golden/tx_2x2_1024_wrappers.bin: for every proof of cloak_2x2_1024.bin the part of a signed ZkVM
payment transaction that surrounds it (header, program, signature), made by the oracle's transaction builder
(oracle/zkvm_tx.c, zko_tx_wrap_payment).  tests/gpu_util.py: load_tx_fixture() puts the transactions back together.
Run in the build container:  python tests/golden/gen_tx_fixture.py"""
import hashlib
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle.binding as oracle  # noqa: E402
from gpu_util import load_cloak_fixture  # noqa: E402

recs, n_in, n_out, plen = load_cloak_fixture()
out = bytearray(b"ZKVMTXW1" + struct.pack("<IIII", len(recs), n_in, n_out, plen))
for i, (com, proof) in enumerate(recs):
    tx = oracle.tx_wrap_payment(n_in, n_out, com, proof, hashlib.sha256(b"tx fixture %d" % i).digest(), mintime=1000 + i, maxtime=10 ** 12)
    assert tx and tx.endswith(struct.pack("<I", plen) + proof)
    wrapper = tx[: len(tx) - 4 - plen]
    if i < 8:
        assert oracle.tx_verify(tx, bytes(64)) == 0
    out += struct.pack("<I", len(wrapper)) + wrapper
open(os.path.join(HERE, "tx_2x2_1024_wrappers.bin"), "wb").write(out)
print(len(out), "bytes,", len(recs), "transactions")
