"""Big-integer restatement of the arithmetic on the ZkVM verification hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under zkvm_amd/ may import this module; it is
used by tests/, by the fixture generators under tests/golden/ and by
oracle/gen_constants.py.  It is exact by construction (Python ints, pow(x,-1,p))
and deliberately slow; the C oracle (oracle/*.c) is the checker that runs at
speed, and this file is what the C oracle is itself checked against.

PARITY UNPINNED vs interstellar/zkvm: /root/reference holds no source (SURVEY.md
section 0), so every function cites the public specification it follows:

  * field / group / encoding .. RFC 9496 (ristretto255), sections 4.1-4.4
  * curve ...................... RFC 7748 section 4.1, RFC 8032 section 5.1
  * Keccak-f[1600], SHA3, SHAKE. FIPS 202
  * STROBE-128 ................. STROBE v1.0.2 (strobe.sourceforge.io/specs)
  * Merlin transcripts ......... merlin.cool (transcript protocol v1.0)
  * Bulletproofs R1CS .......... Bunz et al. 2018 + dalek "r1cs" notes

What *is* pinned: ristretto255 against libsodium 1.0.18 (independent
implementation, this container only -> tests/golden/ristretto255.json), Keccak
against hashlib, Merlin against the public "test protocol" known answer.
"""
from __future__ import annotations

import hashlib
from typing import Iterable, List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------
# GF(2^255-19)                                             RFC 9496 sec 4.1
# --------------------------------------------------------------------------
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493  # group order, sec 4.4
D = (-121665 * pow(121666, -1, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = ((D - 1) * (D - 1)) % P


def is_neg(x: int) -> bool:
    """IS_NEGATIVE: least significant bit of the canonical encoding."""
    return (x % P) & 1 == 1


def ct_abs(x: int) -> int:
    x %= P
    return P - x if is_neg(x) else x


def sqrt_ratio_m1(u: int, v: int) -> Tuple[bool, int]:
    """SQRT_RATIO_M1(u, v), RFC 9496 sec 4.2."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = ct_abs(r)
    return (correct or flipped), r


_ok, _r = sqrt_ratio_m1(1, (-1 - D) % P)
assert _ok
INVSQRT_A_MINUS_D = _r                      # 1/sqrt(a-d), a = -1
# sqrt(a*d-1): RFC 9496 fixes the *odd* root (the other sign changes MAP's output,
# which libsodium's from_hash pins in tests/golden/ristretto255.json).
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
assert SQRT_AD_MINUS_ONE * SQRT_AD_MINUS_ONE % P == (-D - 1) % P
assert INVSQRT_A_MINUS_D == 54469307008909316920995813868745141605393597292927456921205312896311721017578

# RFC 9496 sec 4.1 lists these constants in decimal; the derivations above must
# land on the same representatives (sign matters for ENCODE / MAP).
assert D == 37095705934669439343138083508754565189542113879843219016388785533085940283555
assert SQRT_M1 == 19681161376707505956807079304988542015446066515923890162744021073123829784752
assert ONE_MINUS_D_SQ == 1159843021668779879193775521855586647937357759715417654439879720876111806838
assert D_MINUS_ONE_SQ == 40440834346308536858101042469323190826248399146238708352240133220865137265952

# --------------------------------------------------------------------------
# Edwards25519 points in extended coordinates (X:Y:Z:T), a = -1
# --------------------------------------------------------------------------
Point = Tuple[int, int, int, int]
IDENTITY: Point = (0, 1, 1, 0)

# RFC 8032 sec 5.1 base point
_BY = 4 * pow(5, -1, P) % P
_BX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
BASE: Point = (_BX, _BY, 1, _BX * _BY % P)
assert (-_BX * _BX + _BY * _BY - 1 - D * _BX * _BX % P * _BY * _BY) % P == 0


def pt_add(p: Point, q: Point) -> Point:
    """add-2008-hwcd-3 for a = -1 (Hisil-Wong-Carter-Dawson 2008, sec 3.1)."""
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P
    b = (y1 + x1) * (y2 + x2) % P
    c = t1 * 2 * D % P * t2 % P
    d = z1 * 2 * z2 % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def pt_double(p: Point) -> Point:
    return pt_add(p, p)


def pt_neg(p: Point) -> Point:
    x, y, z, t = p
    return ((-x) % P, y, z, (-t) % P)


def pt_mul(k: int, p: Point) -> Point:
    k %= L
    acc = IDENTITY
    while k:
        if k & 1:
            acc = pt_add(acc, p)
        p = pt_double(p)
        k >>= 1
    return acc


def pt_eq(p: Point, q: Point) -> bool:
    """Ristretto equality, RFC 9496 sec 4.3.3."""
    x1, y1, _, _ = p
    x2, y2, _, _ = q
    return (x1 * y2 - y1 * x2) % P == 0 or (y1 * y2 - x1 * x2) % P == 0


def pt_is_identity(p: Point) -> bool:
    return pt_eq(p, IDENTITY)


# --------------------------------------------------------------------------
# ristretto255 encoding                                   RFC 9496 sec 4.3
# --------------------------------------------------------------------------
def decode(b: bytes) -> Optional[Point]:
    """DECODE, sec 4.3.1.  Returns None for every rejected encoding."""
    if len(b) != 32:
        return None
    s = int.from_bytes(b, "little")
    if s >= P or s & 1:
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = ct_abs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def encode(p: Point) -> bytes:
    """ENCODE, sec 4.3.2."""
    x0, y0, z0, t0 = p
    u1 = (z0 + y0) * (z0 - y0) % P
    u2 = x0 * y0 % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * t0 % P
    ix0 = x0 * SQRT_M1 % P
    iy0 = y0 * SQRT_M1 % P
    enchanted = den1 * INVSQRT_A_MINUS_D % P
    rotate = is_neg(t0 * z_inv % P)
    if rotate:
        x, y, den_inv = iy0, ix0, enchanted
    else:
        x, y, den_inv = x0, y0, den2
    if is_neg(x * z_inv % P):
        y = (-y) % P
    s = ct_abs(den_inv * ((z0 - y) % P) % P)
    return s.to_bytes(32, "little")


def elligator_map(t: int) -> Point:
    """MAP, sec 4.3.4."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-ct_abs(s * t % P)) % P
    if not was_square:
        s = s_prime
    c = (-1) % P if was_square else r
    n = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = n * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


def from_uniform_bytes(b: bytes) -> Point:
    """Element derivation from 64 uniform bytes, sec 4.3.4."""
    assert len(b) == 64
    t1 = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    t2 = int.from_bytes(b[32:], "little") & ((1 << 255) - 1)
    return pt_add(elligator_map(t1 % P), elligator_map(t2 % P))


def msm(scalars: Sequence[int], points: Sequence[Point]) -> Point:
    acc = IDENTITY
    for k, p in zip(scalars, points):
        acc = pt_add(acc, pt_mul(k, p))
    return acc


# --------------------------------------------------------------------------
# Scalars                                                  RFC 9496 sec 4.4
# --------------------------------------------------------------------------
def sc_from_wide(b: bytes) -> int:
    return int.from_bytes(b, "little") % L


def sc_to_bytes(x: int) -> bytes:
    return (x % L).to_bytes(32, "little")


def sc_from_canonical(b: bytes) -> Optional[int]:
    x = int.from_bytes(b, "little")
    return x if x < L else None


# --------------------------------------------------------------------------
# Keccak-f[1600]                                                   FIPS 202
# --------------------------------------------------------------------------
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x: int, n: int) -> int:
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray) -> None:
    a = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def _sponge(rate: int, suffix: int, data: bytes, outlen: int) -> bytes:
    st = bytearray(200)
    pos = 0
    for byte in data:
        st[pos] ^= byte
        pos += 1
        if pos == rate:
            keccak_f1600(st)
            pos = 0
    st[pos] ^= suffix
    st[rate - 1] ^= 0x80
    keccak_f1600(st)
    out = bytearray()
    while len(out) < outlen:
        out += st[:rate]
        if len(out) < outlen:
            keccak_f1600(st)
    return bytes(out[:outlen])


def sha3_512(data: bytes) -> bytes:
    return _sponge(72, 0x06, data, 64)


def shake256(data: bytes, outlen: int) -> bytes:
    return _sponge(136, 0x1F, data, outlen)


# --------------------------------------------------------------------------
# STROBE-128 (the subset Merlin uses) and Merlin v1.0 transcripts
# --------------------------------------------------------------------------
_FLAG_I, _FLAG_A, _FLAG_C, _FLAG_T, _FLAG_M, _FLAG_K = 1, 2, 4, 8, 16, 32
_STROBE_R = 166


class Strobe128:
    """STROBE v1.0.2 sec 5-6 at security level 128 (rate 166), AD/meta-AD/PRF/KEY."""

    def __init__(self, protocol_label: bytes) -> None:
        st = bytearray(200)
        st[0:6] = bytes([1, _STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.st = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def _run_f(self) -> None:
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[_STROBE_R + 1] ^= 0x80
        keccak_f1600(self.st)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes) -> None:
        for byte in data:
            self.st[self.pos] ^= byte
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()

    def _overwrite(self, data: bytes) -> None:
        for byte in data:
            self.st[self.pos] = byte
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool) -> None:
        if more:
            assert flags == self.cur_flags
            return
        assert flags & _FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (_FLAG_C | _FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool) -> None:
        self._begin_op(_FLAG_M | _FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool) -> None:
        self._begin_op(_FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool = False) -> bytes:
        self._begin_op(_FLAG_I | _FLAG_A | _FLAG_C, more)
        return self._squeeze(n)

    def key(self, data: bytes, more: bool = False) -> None:
        self._begin_op(_FLAG_A | _FLAG_C, more)
        self._overwrite(data)


class Transcript:
    """Merlin transcript (merlin.cool, "Transcript protocol")."""

    def __init__(self, label: bytes) -> None:
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes) -> None:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def append_u64(self, label: bytes, x: int) -> None:
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n)

    # Bulletproofs "TranscriptProtocol" helpers (dalek bulletproofs notes)
    def append_scalar(self, label: bytes, s: int) -> None:
        self.append_message(label, sc_to_bytes(s))

    def append_point(self, label: bytes, p: bytes) -> None:
        self.append_message(label, p)

    def challenge_scalar(self, label: bytes) -> int:
        return sc_from_wide(self.challenge_bytes(label, 64))


# --------------------------------------------------------------------------
# Bulletproofs generators (dalek bulletproofs "generators" notes)
# --------------------------------------------------------------------------
def pedersen_gens() -> Tuple[Point, Point]:
    b = BASE
    b_blinding = from_uniform_bytes(hashlib.sha3_512(encode(BASE)).digest())
    return b, b_blinding


def generators_chain(label: bytes, n: int) -> List[Point]:
    stream = hashlib.shake_256(b"GeneratorsChain" + label).digest(64 * n)
    return [from_uniform_bytes(stream[64 * i: 64 * i + 64]) for i in range(n)]


def bulletproof_gens(n: int, party: int = 0) -> Tuple[List[Point], List[Point]]:
    idx = party.to_bytes(4, "little")
    return generators_chain(b"G" + idx, n), generators_chain(b"H" + idx, n)
