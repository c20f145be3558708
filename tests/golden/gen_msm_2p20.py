#!/usr/bin/env python3
"""Generate tests/golden/msm_2p20.json: the expected value of BASELINE.json configs[2], the 2^20-term
Ristretto255 multiscalar multiplication ("expected output fixture computed
once by CPU restatement").

Inputs are reproducible on both sides from SHAKE256 (seed 0x5a6b564d = "ZkVM"):
    point  i = from_uniform_bytes(SHAKE256(seed_le32 || "msm2p20")[64 i : 64 i + 64])      (RFC 9496 4.3.4)
    scalar i = SHAKE256(seed_le32 || "msm2p20 scalars")[64 i : 64 i + 64] mod l            (wide reduction)
The oracle (oracle/ristretto.c, oracle/msm.c) maps the points and evaluates the sum; the file records the
32-byte result for the full size and for the prefixes 2^16 and 2^18, and SHA-256 digests of the two input
arrays so that the GPU-side regeneration (hash_to_points on the device) is itself checked at full size.

Run:  python tests/golden/gen_msm_2p20.py        (about two minutes on one core)
"""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import binding as oracle  # noqa: E402

SEED = 0x5A6B564D
L = 2**252 + 27742317777372353535851937790883648493
N = 1 << 20


def shake(tag: bytes, n: int) -> bytes:
    return hashlib.shake_256(SEED.to_bytes(4, "little") + tag).digest(n)


def inputs(n: int):
    raw_p = shake(b"msm2p20", 64 * n)
    raw_s = shake(b"msm2p20 scalars", 64 * n)
    lib = oracle.load()
    pts = C.create_string_buffer(32 * n)
    ge = oracle.Ge()
    for i in range(n):
        lib.ristretto_from_uniform_bytes(C.byref(ge), raw_p[64 * i: 64 * i + 64])
        lib.ristretto_encode(C.cast(C.byref(pts, 32 * i), C.c_char_p), C.byref(ge))
    sc = b"".join((int.from_bytes(raw_s[64 * i: 64 * i + 64], "little") % L).to_bytes(32, "little") for i in range(n))
    return sc, pts.raw


def main():
    sc, pt = inputs(N)
    out = {"seed": SEED, "n": N, "point_tag": "msm2p20", "scalar_tag": "msm2p20 scalars",
           "scalars_sha256": hashlib.sha256(sc).hexdigest(), "points_sha256": hashlib.sha256(pt).hexdigest(),
           "results": {}}
    for n in (1 << 16, 1 << 18, N):
        rc, res, _ = oracle.msm(sc[: 32 * n], pt[: 32 * n])
        assert rc == 0
        out["results"][str(n)] = res.hex()
        print(n, res.hex())
    path = os.path.join(HERE, "msm_2p20.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
