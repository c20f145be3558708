#!/usr/bin/env python3
"""Generate tests/golden/cloak_2x2_proofs.bin: 64 R1CS proofs of the 2-in/2-out cloak statement
made by the oracle prover (oracle/r1cs.c, oracle/cloak.c) with seeded witnesses.

The file is DATA -- commitments and proof bytes -- used (a) by bench.py as the real-proof workload
(each proof is verified many times under distinct verifier randomness r) and (b) by the tests as a
frozen input whose verdict must not change.  Layout (little endian):
    8 s  magic "ZKCLOAK1"
    u32  count, u32 n_in, u32 n_out, u32 proof_len
    count x (64 * (n_in + n_out) bytes of commitments, proof_len bytes of proof)

Run:  python tests/golden/gen_cloak_proofs.py
"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import binding as oracle  # noqa: E402

COUNT, N_IN, N_OUT = 64, 2, 2
SEED = b"zkvm_amd cloak fixture v1".ljust(32, b"\0")


def main():
    com, proofs = oracle.cloak_prove_batch(COUNT, N_IN, N_OUT, SEED, threads=8)
    w = 64 * (N_IN + N_OUT)
    plen = len(proofs[0])
    assert all(len(p) == plen for p in proofs)
    for i, p in enumerate(proofs):
        assert oracle.cloak_verify(com[w * i: w * (i + 1)], N_IN, N_OUT, p, bytes(64))
    path = os.path.join(HERE, "cloak_2x2_proofs.bin")
    with open(path, "wb") as f:
        f.write(b"ZKCLOAK1" + struct.pack("<IIII", COUNT, N_IN, N_OUT, plen))
        for i, p in enumerate(proofs):
            f.write(com[w * i: w * (i + 1)] + p)
    print("wrote", path, os.path.getsize(path), "bytes; proof_len", plen)


if __name__ == "__main__":
    main()
