#!/usr/bin/env python3
"""Generate tests/golden/ristretto255.json with libsodium 1.0.18.

libsodium is an implementation of ristretto255 that shares no code with this
repository (nor with curve25519-dalek); it exists in the build container at
/opt/conda/lib/libsodium.so.23 and is used ONLY here, to emit data.  The JSON
file is what travels: inputs and expected outputs, no code.

Known libsodium 1.0.18 deviation, handled below: `is_valid_point` ignores bit
255 of the encoding, whereas RFC 9496 sec 4.3.1 (and dalek) reject any s >= p.
Encodings with bit 255 set are therefore recorded under "rfc_only_reject" with
the RFC's verdict and are not attributed to libsodium.

Run:  python tests/golden/gen_ristretto255.py
"""
import ctypes
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
so = ctypes.CDLL("/opt/conda/lib/libsodium.so.23")
assert so.sodium_init() >= 0

L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19


def stream(tag: bytes, n: int) -> bytes:
    return hashlib.shake_256(b"zkvm_amd golden v1|" + tag).digest(n)


def buf():
    return ctypes.create_string_buffer(32)


def base_mul(k: int) -> bytes:
    out = buf()
    rc = so.crypto_scalarmult_ristretto255_base(out, (k % L).to_bytes(32, "little"))
    return bytes(32) if rc != 0 else out.raw  # rc=-1 <=> identity (libsodium quirk)


def mul(k: int, p: bytes) -> bytes:
    out = buf()
    rc = so.crypto_scalarmult_ristretto255(out, (k % L).to_bytes(32, "little"), p)
    return bytes(32) if rc != 0 else out.raw


def add(p: bytes, q: bytes) -> bytes:
    out = buf()
    assert so.crypto_core_ristretto255_add(out, p, q) == 0
    return out.raw


def from_hash(h: bytes) -> bytes:
    out = buf()
    assert so.crypto_core_ristretto255_from_hash(out, h) == 0
    return out.raw


def scalar_op(name: str, a: int, b: int) -> int:
    out = buf()
    getattr(so, "crypto_core_ristretto255_scalar_" + name)(out, a.to_bytes(32, "little"), b.to_bytes(32, "little"))
    return int.from_bytes(out.raw, "little")


def main():
    g = {"source": "libsodium 1.0.18 (crypto_core_ristretto255_*, crypto_scalarmult_ristretto255*)"}

    # 1. small multiples of the generator (RFC 9496 appendix A.1 lists 0..15)
    g["base_multiples"] = [base_mul(k).hex() for k in range(17)]

    # 2. random scalar * B, and k * P for derived P
    sm = []
    for i in range(24):
        k = int.from_bytes(stream(b"k%d" % i, 64), "little") % L
        p = from_hash(stream(b"p%d" % i, 64))
        sm.append({"k": "%064x" % k, "kB": base_mul(k).hex(), "P": p.hex(), "kP": mul(k, p).hex()})
    # edge scalars
    for k in (0, 1, 2, L - 1, L - 2, 2**252, (L - 1) // 2):
        p = from_hash(stream(b"edge", 64))
        sm.append({"k": "%064x" % k, "kB": base_mul(k).hex(), "P": p.hex(), "kP": mul(k, p).hex()})
    g["scalarmult"] = sm

    # 3. addition
    ad = []
    for i in range(16):
        p, q = from_hash(stream(b"a%d" % i, 64)), from_hash(stream(b"b%d" % i, 64))
        ad.append({"P": p.hex(), "Q": q.hex(), "sum": add(p, q).hex(), "dbl": add(p, p).hex()})
    g["add"] = ad

    # 4. element derivation from 64 uniform bytes (RFC 9496 sec 4.3.4)
    fh = []
    for i in range(32):
        h = stream(b"h%d" % i, 64)
        fh.append({"in": h.hex(), "out": from_hash(h).hex()})
    for h in (bytes(64), b"\xff" * 64, b"\x01" + bytes(63), bytes(32) + b"\x01" + bytes(31)):
        fh.append({"in": h.hex(), "out": from_hash(h).hex()})
    g["from_uniform_bytes"] = fh

    # 5. validity of encodings, bit 255 clear (libsodium and the RFC agree)
    val = []
    raw = stream(b"enc", 32 * 400)
    for i in range(400):
        b = bytearray(raw[32 * i: 32 * i + 32])
        b[31] &= 0x7F
        b = bytes(b)
        val.append({"enc": b.hex(), "valid": int(so.crypto_core_ristretto255_is_valid_point(b) == 1)})
    # structured encodings: 0, 1, p-1, p, p+1 (non-canonical), small evens/odds
    for s in [0, 1, 2, 3, 4, P - 1, P - 2, 2**254, 2**255 - 20]:
        b = s.to_bytes(32, "little")
        val.append({"enc": b.hex(), "valid": int(so.crypto_core_ristretto255_is_valid_point(b) == 1)})
    g["valid_encoding"] = val
    # non-canonical field encodings p .. p+18 fit in 255 bits: both reject
    g["noncanonical"] = [{"enc": (P + d).to_bytes(32, "little").hex(),
                          "valid": int(so.crypto_core_ristretto255_is_valid_point((P + d).to_bytes(32, "little")) == 1)}
                         for d in range(0, 19)]
    # bit 255 set: RFC 9496 rejects (s >= 2^255 > p); libsodium 1.0.18 masks the bit
    g["rfc_only_reject"] = [(int.from_bytes(bytes.fromhex(v["enc"]), "little") | (1 << 255)).to_bytes(32, "little").hex()
                            for v in val[:40] if v["valid"]]

    # 6. scalar field
    scs = []
    for i in range(24):
        a = int.from_bytes(stream(b"sa%d" % i, 64), "little") % L
        b = int.from_bytes(stream(b"sb%d" % i, 64), "little") % L
        inv = buf()
        so.crypto_core_ristretto255_scalar_invert(inv, a.to_bytes(32, "little"))
        wide = stream(b"sw%d" % i, 64)
        red = buf()
        so.crypto_core_ristretto255_scalar_reduce(red, wide)
        scs.append({"a": "%064x" % a, "b": "%064x" % b,
                    "add": "%064x" % scalar_op("add", a, b), "sub": "%064x" % scalar_op("sub", a, b),
                    "mul": "%064x" % scalar_op("mul", a, b),
                    "inv_a": "%064x" % int.from_bytes(inv.raw, "little"),
                    "wide": wide.hex(), "wide_reduced": "%064x" % int.from_bytes(red.raw, "little")})
    g["scalars"] = scs

    # 7. small multiscalar multiplications  sum k_i P_i  (naive, via libsodium)
    ms = []
    for n in (1, 2, 3, 17, 64, 200):
        ks, ps = [], []
        acc = None
        for i in range(n):
            k = int.from_bytes(stream(b"mk%d.%d" % (n, i), 64), "little") % L
            p = from_hash(stream(b"mp%d.%d" % (n, i), 64))
            ks.append(k)
            ps.append(p)
            t = mul(k, p)
            acc = t if acc is None else add(acc, t)
        ms.append({"scalars": b"".join(k.to_bytes(32, "little") for k in ks).hex(),
                   "points": b"".join(ps).hex(), "result": acc.hex()})
    # an MSM that sums to the identity: k*P + (l-k)*P
    k = int.from_bytes(stream(b"zk", 64), "little") % L
    p = from_hash(stream(b"zp", 64))
    ms.append({"scalars": (k.to_bytes(32, "little") + (L - k).to_bytes(32, "little")).hex(),
               "points": (p + p).hex(), "result": bytes(32).hex()})
    g["msm"] = ms

    with open(os.path.join(HERE, "ristretto255.json"), "w") as f:
        json.dump(g, f, indent=0, sort_keys=True)
    print("wrote ristretto255.json:", {k: (len(v) if isinstance(v, list) else v) for k, v in g.items()})


if __name__ == "__main__":
    main()
