"""C oracle field + scalar layers vs Python big ints (exact by construction).
Reference rows: SURVEY.md sec 8(a) a1-a3 (FieldElement51, invert/sqrt_ratio_i, Scalar)."""
import random

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493

EDGE_F = [0, 1, 2, 19, P - 1, P - 2, P - 19, (P - 1) // 2, (P + 1) // 2, 2**51 - 1, 2**51, 2**102, 2**204 - 1,
          2**254, 2**255 - 20]


def _samples(rng, n):
    return EDGE_F + [rng.randrange(P) for _ in range(n)]


def test_fe_roundtrip_and_canonical(oracle):
    rng = random.Random(1)
    for x in _samples(rng, 200):
        assert oracle.fe_to_int(oracle.fe_from_int(x)) == x % P
    # frombytes ignores bit 255 and reduces p..2^255-1
    for x in [P, P + 1, P + 18, 2**255 - 1]:
        assert oracle.fe_to_int(oracle.fe_from_int(x)) == x % P


def test_fe_ring_ops(oracle):
    rng = random.Random(2)
    xs = _samples(rng, 60)
    for a in xs:
        for b in rng.sample(xs, 6):
            assert oracle.fe_binop("fe_add", a, b) == (a + b) % P
            assert oracle.fe_binop("fe_sub", a, b) == (a - b) % P
            assert oracle.fe_binop("fe_mul", a, b) == (a * b) % P
        assert oracle.fe_unop("fe_sq", a) == a * a % P
        assert oracle.fe_unop("fe_neg", a) == (-a) % P


def test_fe_invert_pow(oracle):
    rng = random.Random(3)
    for a in _samples(rng, 20):
        assert oracle.fe_unop("fe_pow22523", a) == pow(a, (P - 5) // 8, P)
        if a % P:
            assert oracle.fe_unop("fe_invert", a) == pow(a, -1, P)
    assert oracle.fe_unop("fe_invert", 0) == 0


def test_sqrt_ratio_m1(oracle, pyref):
    rng = random.Random(4)
    n_sq = 0
    for _ in range(60):
        u, v = rng.randrange(P), rng.randrange(1, P)
        ok, r = oracle.fe_sqrt_ratio_m1(u, v)
        ok_ref, r_ref = pyref.sqrt_ratio_m1(u, v)
        assert (ok, r) == (ok_ref, r_ref)
        n_sq += ok
    assert 10 < n_sq < 50
    # u = 0 -> (True, 0);  v = 0, u != 0 -> (False, 0)
    assert oracle.fe_sqrt_ratio_m1(0, 5) == (True, 0)
    assert oracle.fe_sqrt_ratio_m1(7, 0) == (False, 0)


def test_scalar_ops_vs_bigint(oracle):
    rng = random.Random(5)
    edge = [0, 1, 2, L - 1, L - 2, 2**252, 2**252 - 1, (L - 1) // 2]
    xs = edge + [rng.randrange(L) for _ in range(60)]
    for a in xs:
        for b in rng.sample(xs, 5):
            assert oracle.sc_binop("sc_add", a, b) == (a + b) % L
            assert oracle.sc_binop("sc_sub", a, b) == (a - b) % L
            assert oracle.sc_binop("sc_mul", a, b) == (a * b) % L
    for a in xs[:12]:
        if a:
            assert oracle.sc_invert(a) == pow(a, -1, L)


def test_scalar_wide_reduction(oracle):
    rng = random.Random(6)
    for x in [0, L, L - 1, 2**512 - 1, 2**256, 2**504, L * L, 2**511 + 12345]:
        assert oracle.sc_reduce_wide((x % 2**512).to_bytes(64, "little")) == x % 2**512 % L
    for _ in range(300):
        x = rng.getrandbits(512)
        assert oracle.sc_reduce_wide(x.to_bytes(64, "little")) == x % L


def test_scalar_golden_libsodium(oracle, golden):
    for v in golden["scalars"]:
        a, b = int(v["a"], 16), int(v["b"], 16)
        assert oracle.sc_binop("sc_add", a, b) == int(v["add"], 16)
        assert oracle.sc_binop("sc_sub", a, b) == int(v["sub"], 16)
        assert oracle.sc_binop("sc_mul", a, b) == int(v["mul"], 16)
        assert oracle.sc_invert(a) == int(v["inv_a"], 16)
        assert oracle.sc_reduce_wide(bytes.fromhex(v["wide"])) == int(v["wide_reduced"], 16)
