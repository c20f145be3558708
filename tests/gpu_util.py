"""Shared helpers for the GPU parity tests (seeded inputs, bitmap decoding)."""
import hashlib

L = 2**252 + 27742317777372353535851937790883648493
BAD_POINT = bytes.fromhex("01" + "00" * 31)          # s = 1 is negative: rejected by RFC 9496 DECODE


def stream(tag: str, n: int) -> bytes:
    return hashlib.shake_256(b"zkvm_amd test|" + tag.encode()).digest(n)


def scalars(tag: str, n: int) -> bytes:
    raw = stream("sc|" + tag, 64 * n)
    return b"".join((int.from_bytes(raw[64 * i: 64 * i + 64], "little") % L).to_bytes(32, "little") for i in range(n))


def points(oracle, tag: str, n: int, distinct: int = 0) -> bytes:
    """n valid encodings; `distinct` > 0 cycles through that many (cheap for big n)."""
    d = distinct or n
    raw = stream("pt|" + tag, 64 * d)
    uniq = [oracle.from_uniform_bytes(raw[64 * i: 64 * i + 64]) for i in range(d)]
    return b"".join(uniq[i % d] for i in range(n))


def bits(bm: bytes, n: int):
    return [(bm[i // 8] >> (i % 8)) & 1 for i in range(n)]
