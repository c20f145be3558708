#!/usr/bin/env python3
"""Generate the round-2 proof fixtures with the oracle prover (oracle/r1cs.c, oracle/cloak.c):

  cloak_2x2_1024.bin   1024 DISTINCT R1CS proofs of the 2-in/2-out cloak statement (bench.py's headline
                       workload, BASELINE.json configs[1], and the full-size parity test): format of
                       gen_cloak_proofs.py ("ZKCLOAK1", count, n_in, n_out, proof_len, records).
  cloak_mixed.bin      32 proofs of each of the shapes of BASELINE.json configs[3] / SURVEY.md sec 8(d)
                       config 4 -- 1x1, 1x2, 2x2, 3x3, 4x4 --:
                           8 s  magic "ZKCLOAKM", u32 n_groups,
                           per group: u32 count, n_in, n_out, proof_len, then count x (commitments, proof)

The files are DATA (commitments and proof bytes).  Run:  python tests/golden/gen_cloak_fixtures_r2.py
"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import binding as oracle  # noqa: E402

SHAPES = [(1, 1), (1, 2), (2, 2), (3, 3), (4, 4)]


def prove(count, n_in, n_out, seed):
    com, proofs = oracle.cloak_prove_batch(count, n_in, n_out, seed.ljust(32, b"\0"), threads=8)
    w = 64 * (n_in + n_out)
    plen = len(proofs[0])
    assert all(len(p) == plen for p in proofs)
    acc = oracle.cloak_verify_batch(com, n_in, n_out, b"".join(proofs), plen, bytes(64 * count), threads=8)
    assert acc == b"\x01" * count
    return [(com[w * i: w * (i + 1)], p) for i, p in enumerate(proofs)], plen


def main():
    recs, plen = prove(1024, 2, 2, b"zkvm_amd cloak fixture r2 2x2")
    assert len({p for _, p in recs}) == 1024
    path = os.path.join(HERE, "cloak_2x2_1024.bin")
    with open(path, "wb") as f:
        f.write(b"ZKCLOAK1" + struct.pack("<IIII", len(recs), 2, 2, plen))
        for com, p in recs:
            f.write(com + p)
    print("wrote", path, os.path.getsize(path), "bytes; proof_len", plen)
    path = os.path.join(HERE, "cloak_mixed.bin")
    with open(path, "wb") as f:
        f.write(b"ZKCLOAKM" + struct.pack("<I", len(SHAPES)))
        for n_in, n_out in SHAPES:
            recs, plen = prove(32, n_in, n_out, b"zkvm_amd cloak fixture r2 %dx%d" % (n_in, n_out))
            f.write(struct.pack("<IIII", len(recs), n_in, n_out, plen))
            for com, p in recs:
                f.write(com + p)
            print("  shape %dx%d: proof_len %d" % (n_in, n_out, plen))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
