"""The arithmetic layers of the device code on their own, against Python integers:
GF(2^255-19) on 26+25-bit limb pairs (field.hpp) and the integers mod l in both device forms (sc_dev.hpp: canonical
Montgomery words, lazy ten-limb form), including the edge values of every representation."""
import random

import pytest

pytestmark = pytest.mark.gpu
P = 2 ** 255 - 19
L = 2 ** 252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def ctx():
    from zkvm_amd import Context
    c = Context(0)
    yield c
    c.close()


def _vals(rng, mod, n):
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, 2 ** 255 - 1, 2 ** 256 - 1, 2 ** 252, 2 ** 252 - 1, mod, mod + 1, 19, 2 ** 51 - 1,
            2 ** 26 - 1, 2 ** 26, (1 << 255) - 20, sum(((1 << 26) - 1) << (26 * i) for i in range(10)) % 2 ** 256]
    out = [e % 2 ** 256 for e in edge]
    while len(out) < n:
        out.append(rng.getrandbits(256))
    return out[:n]


def _run(ctx, op, xs, ys):
    a = b"".join(x.to_bytes(32, "little") for x in xs)
    b = b"".join(y.to_bytes(32, "little") for y in ys)
    raw = ctx.debug_arith(op, a, b)
    return [int.from_bytes(raw[32 * i: 32 * i + 32], "little") for i in range(len(xs))]


def test_field_arithmetic_vs_python_integers(ctx):
    rng = random.Random(255)
    n = 4096
    xs, ys = _vals(rng, P, n), list(reversed(_vals(rng, P, n)))
    fx = [(x % 2 ** 255) % P for x in xs]                       # bit 255 is ignored on input
    fy = [(y % 2 ** 255) % P for y in ys]
    assert _run(ctx, 0, xs, ys) == [a * b % P for a, b in zip(fx, fy)]
    assert _run(ctx, 1, xs, ys) == [a * a % P for a in fx]
    assert _run(ctx, 2, xs, ys) == [pow(a, P - 2, P) for a in fx]
    assert _run(ctx, 3, xs, ys) == [2 * a % P for a in fx]
    assert _run(ctx, 4, xs, ys) == [pow(a, (P - 5) // 8, P) for a in fx]


def test_scalar_arithmetic_in_both_forms_vs_python_integers(ctx):
    rng = random.Random(252)
    n = 4096
    xs, ys = _vals(rng, L, n), list(reversed(_vals(rng, L, n)))
    sx, sy = [x % L for x in xs], [y % L for y in ys]
    want_mul = [a * b % L for a, b in zip(sx, sy)]
    assert _run(ctx, 10, xs, ys) == want_mul
    assert _run(ctx, 11, xs, ys) == want_mul
    assert _run(ctx, 12, xs, ys) == [((a - b) * (a + b) + 16 * a * b - b) % L for a, b in zip(sx, sy)]
    want_inv = [pow(a, L - 2, L) for a in sx]
    assert _run(ctx, 13, xs[:512], ys[:512]) == want_inv[:512]
    assert _run(ctx, 14, xs[:512], ys[:512]) == want_inv[:512]
    assert _run(ctx, 16, xs[:512], ys[:512]) == want_inv[:512]          # the fixed chain of the lane-per-proof kernels
    assert _run(ctx, 16, [0, 1, L - 1], [0, 0, 0]) == [0, 1, L - 1]
    assert _run(ctx, 15, xs, ys) == sy
