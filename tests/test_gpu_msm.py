"""GPU parity: libzkgpu (HIP, through the C ABI) vs the CPU oracle and the libsodium golden
vectors.  Bit-exact on the 32-byte canonical encodings / accept bitmaps.
Reference rows: SURVEY.md sec 8(a) a4-a7, a9 (tail)."""
import random

import pytest

from gpu_util import BAD_POINT, L, bits, points, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from zkvm_amd import Context
    c = Context(0)
    yield c
    c.close()


def test_decode_check_golden(ctx, golden):
    encs = [bytes.fromhex(v["enc"]) for v in golden["valid_encoding"] + golden["noncanonical"]]
    want = [v["valid"] for v in golden["valid_encoding"] + golden["noncanonical"]]
    encs += [bytes.fromhex(h) for h in golden["rfc_only_reject"]]
    want += [0] * len(golden["rfc_only_reject"])
    assert list(ctx.decode_check(b"".join(encs))) == want


def test_msm_golden_libsodium(ctx, golden):
    for v in golden["msm"]:
        out = ctx.msm(bytes.fromhex(v["scalars"]), bytes.fromhex(v["points"]))
        assert out.hex() == v["result"]
    # scalar multiplication vectors as 1-term MSMs
    for v in golden["scalarmult"]:
        k = int(v["k"], 16).to_bytes(32, "little")
        assert ctx.msm(k, bytes.fromhex(v["P"])).hex() == v["kP"]
    # from_uniform_bytes outputs decode and re-encode (1 * P)
    one = (1).to_bytes(32, "little")
    for v in golden["from_uniform_bytes"][:8]:
        assert ctx.msm(one, bytes.fromhex(v["out"])).hex() == v["out"]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 255, 1000, 4097])
def test_msm_vs_oracle_sizes(ctx, oracle, n):
    sc, pt = scalars("sz%d" % n, n), points(oracle, "sz%d" % n, n, distinct=min(n, 50))
    rc, want, _ = oracle.msm(sc, pt)
    assert rc == 0
    assert ctx.msm(sc, pt) == want


@pytest.mark.parametrize("w", [4, 5, 7, 8, 11, 13, 15, 16])
def test_msm_every_window_width(ctx, oracle, w):
    n = 700
    sc, pt = scalars("w%d" % w, n), points(oracle, "w", n, distinct=40)
    rc, want, _ = oracle.msm(sc, pt)
    ctx.set_window_bits(w)
    try:
        assert ctx.msm(sc, pt) == want
        assert ctx.last_window_bits() == w
    finally:
        ctx.set_window_bits(0)


def test_msm_structured_scalars(ctx, oracle):
    n = 600
    pt = points(oracle, "structured", n, distinct=30)
    same = pt[:32] * n
    for name, k in [("zero", 0), ("one", 1), ("lm1", L - 1), ("2^252", 2**252), ("2^254", 2**254), ("top", 2**255 - 1)]:
        sc = k.to_bytes(32, "little") * n
        for p in (pt, same):
            rc, want, _ = oracle.msm(sc, p)
            assert ctx.msm(sc, p) == want, name
    # everything cancels -> identity (all-zero encoding)
    half = scalars("cancel", n // 2)
    neg = b"".join(((L - int.from_bytes(half[32 * i: 32 * i + 32], "little")) % L).to_bytes(32, "little")
                   for i in range(n // 2))
    assert ctx.msm(half + neg, pt[: 16 * n] + pt[: 16 * n]) == bytes(32)


def test_msm_invalid_point_and_scalar(ctx, oracle):
    from zkvm_amd import ZkGpuError
    n = 300
    sc, pt = scalars("inv", n), bytearray(points(oracle, "inv", n, distinct=20))
    pt[32 * 123: 32 * 124] = BAD_POINT
    pt[32 * 250: 32 * 251] = BAD_POINT
    with pytest.raises(ZkGpuError) as e:
        ctx.msm(sc, bytes(pt))
    assert e.value.code == -2 and e.value.index == 123
    # scalar with bit 255 set violates the `Scalar` invariant -> EINVAL
    bad_sc = bytearray(sc)
    bad_sc[31] |= 0x80
    with pytest.raises(ZkGpuError) as e:
        ctx.msm(bytes(bad_sc), points(oracle, "inv", n, distinct=20))
    assert e.value.code == -1


def test_msm_large_vs_oracle(ctx, oracle):
    n = 1 << 16
    sc, pt = scalars("large", n), points(oracle, "large", n, distinct=257)
    rc, want, _ = oracle.msm(sc, pt)
    assert ctx.msm(sc, pt) == want
    assert ctx.last_window_bits() >= 11


def test_msm_full_size_properties(ctx, oracle):
    """BASELINE config 3 size (2^20 terms): linearity and a closed form instead of the oracle.
    sum_i k_i P  = (sum k_i) P ;  MSM(a) + MSM(b) == MSM(a + b)."""
    n = 1 << 20
    p = points(oracle, "full", 1)
    a, b = scalars("fulla", n), scalars("fullb", n)
    ia = [int.from_bytes(a[32 * i: 32 * i + 32], "little") for i in range(n)]
    ib = [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(n)]
    same = p * n
    ra, rb = ctx.msm(a, same), ctx.msm(b, same)
    assert ra == ctx.msm((sum(ia) % L).to_bytes(32, "little"), p)
    assert rb == ctx.msm((sum(ib) % L).to_bytes(32, "little"), p)
    ab = b"".join(((x + y) % L).to_bytes(32, "little") for x, y in zip(ia, ib))
    rab = ctx.msm(ab, same)
    one = (1).to_bytes(32, "little")
    assert ctx.msm(one + one, ra + rb) == rab
    # distinct points: split the sum in two halves, recombine
    pts = points(oracle, "fullp", n, distinct=4099)
    whole = ctx.msm(a, pts)
    h = n // 2
    lo, hi = ctx.msm(a[: 32 * h], pts[: 32 * h]), ctx.msm(a[32 * h:], pts[32 * h:])
    assert ctx.msm(one + one, lo + hi) == whole


def _make_batch(oracle, rng, sizes, corrupt):
    base = oracle.basepoint()
    sc, pt, offs, want = b"", b"", [0], []
    for i, n in enumerate(sizes):
        terms = []
        for _ in range(n // 2):
            k = rng.randrange(1, L)
            p = oracle.encode(oracle.scalarmult(rng.randrange(1, L), base))
            terms += [(k, p), (L - k, p)]
        good = True
        kind = corrupt.get(i)
        if kind == "scalar" and terms:
            terms[0] = ((terms[0][0] + 1) % L, terms[0][1]); good = False
        if kind == "point" and terms:
            terms[-1] = (terms[-1][0], BAD_POINT); good = False
        rng.shuffle(terms)
        sc += b"".join(k.to_bytes(32, "little") for k, _ in terms)
        pt += b"".join(p for _, p in terms)
        offs.append(offs[-1] + len(terms))
        want.append(int(good))
    return sc, pt, offs, want


def test_verify_batch_ragged_vs_oracle(ctx, oracle):
    rng = random.Random(21)
    sizes = [rng.choice([0, 2, 4, 10, 64, 200, 550]) for _ in range(97)]
    corrupt = {i: ("scalar" if i % 7 == 3 else "point") for i in range(97) if i % 7 in (3, 5)}
    sc, pt, offs, want = _make_batch(oracle, rng, sizes, corrupt)
    want = [w if s else 1 for w, s in zip(want, sizes)]      # empty MSM == identity
    bm = ctx.verify_batch(sc, pt, offs)
    assert bits(bm, len(sizes)) == want
    assert bm == oracle.verify_batch(sc, pt, offs)
    assert ctx.verify_batch(b"", b"", [0]) == b""
    assert ctx.verify_batch(b"", b"", [0, 0, 0]) == b"\x03"


def test_verify_batch_with_pointset_vs_generic(ctx, oracle):
    """static (generator) terms by index + dynamic terms == the same checks through the generic CSR path."""
    from zkvm_amd import PointSet
    rng = random.Random(22)
    n_gen = 64
    gens = points(oracle, "gens", n_gen)
    ps = PointSet(ctx, gens)
    assert len(ps) == n_gen
    dyn_sc, dyn_pt, dyn_off = b"", b"", [0]
    st_sc, st_idx, st_off = b"", [], [0]
    gen_sc, gen_pt, gen_off = b"", b"", [0]
    for i in range(40):
        # static part: random scalars on a random subset of generators
        idx = rng.sample(range(n_gen), rng.choice([1, 8, 64]))
        ks = [rng.randrange(L) for _ in idx]
        # dynamic part: points whose combination cancels the static part  ->  -(sum k_j G_j) as one extra point
        pts = [oracle.decode(gens[32 * j: 32 * j + 32]) for j in idx]
        tot = oracle.msm_points("vartime", ks, pts)
        t_enc = oracle.encode(tot)
        extra_k = (L - 1) if i % 5 else (L - 2)      # every 5th check is wrong
        st_sc += b"".join(k.to_bytes(32, "little") for k in ks)
        st_idx += idx
        st_off.append(st_off[-1] + len(idx))
        dyn_sc += extra_k.to_bytes(32, "little")
        dyn_pt += t_enc
        dyn_off.append(dyn_off[-1] + 1)
        gen_sc += b"".join(k.to_bytes(32, "little") for k in ks) + extra_k.to_bytes(32, "little")
        gen_pt += b"".join(gens[32 * j: 32 * j + 32] for j in idx) + t_enc
        gen_off.append(gen_off[-1] + len(idx) + 1)
    want = oracle.verify_batch(gen_sc, gen_pt, gen_off)
    assert bits(want, 40) == [1 if i % 5 else 0 for i in range(40)]
    assert ctx.verify_batch(gen_sc, gen_pt, gen_off) == want
    assert ctx.verify_batch_ps(ps, dyn_sc, dyn_pt, dyn_off, st_sc, st_off, static_index=st_idx) == want
    ps.close()
    from zkvm_amd import ZkGpuError
    with pytest.raises(ZkGpuError):
        PointSet(ctx, gens[:64] + BAD_POINT)


def test_determinism_two_runs(ctx, oracle):
    n = 5000
    sc, pt = scalars("det", n), points(oracle, "det", n, distinct=100)
    assert ctx.msm(sc, pt) == ctx.msm(sc, pt)


def test_generators_and_hash_to_point_vs_oracle(ctx, oracle, golden):
    """Product-side generator derivation vs the oracle and dalek's published constant."""
    b, bb = ctx.pedersen_gens()
    assert (b, bb) == oracle.pedersen_gens()
    assert bb.hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"
    g, h = ctx.bulletproof_gens(40)
    assert [g[32 * i: 32 * i + 32] for i in range(40)] == oracle.bulletproof_gens(40, "G")
    assert [h[32 * i: 32 * i + 32] for i in range(40)] == oracle.bulletproof_gens(40, "H")
    g1, _ = ctx.bulletproof_gens(3, party=1)
    assert [g1[32 * i: 32 * i + 32] for i in range(3)] == oracle.bulletproof_gens(3, "G", party=1)
    ins = b"".join(bytes.fromhex(v["in"]) for v in golden["from_uniform_bytes"])
    outs = b"".join(bytes.fromhex(v["out"]) for v in golden["from_uniform_bytes"])
    assert ctx.hash_to_points(ins) == outs          # libsodium-generated vectors


def test_msm_batch_values_vs_oracle(ctx, oracle):
    rng = random.Random(31)
    sizes = [0, 1, 2, 5, 33, 190, 549]
    sc, pt, offs = b"", b"", [0]
    want = []
    for i, n in enumerate(sizes):
        s, p = scalars("mb%d" % i, n), bytearray(points(oracle, "mb%d" % i, n, distinct=min(n, 20) or 1)[: 32 * n])
        if i == 3:
            p[32:64] = BAD_POINT
        rc, enc, _ = oracle.msm(s, bytes(p))
        want.append(enc if rc == 0 else bytes(32))
        sc += s
        pt += bytes(p)
        offs.append(offs[-1] + n)
    out, ok = ctx.msm_batch(sc, pt, offs)
    assert [out[32 * i: 32 * i + 32] for i in range(len(sizes))] == want
    assert bits(ok, len(sizes)) == [1, 1, 1, 0, 1, 1, 1]


def test_many_small_rows_uniform_and_with_one_very_long_row(ctx, oracle):
    """zkgpu_msm_batch with >= 64 rows of a few terms takes the per-point-table path (one workgroup per row); a batch whose
    AVERAGE row is short but which holds one row of 3000 terms must not (the longest row is bounded too, such a
    batch goes through the bucket pipeline).  Both against the oracle, value by value."""
    for tag, sizes in (("u", [3] * 40 + [1, 0, 7, 2] * 10), ("s", [2] * 100 + [3000] + [4] * 27)):
        sc, pt, offs, want = b"", b"", [0], []
        for i, n in enumerate(sizes):
            s, p = scalars("%s%d" % (tag, i), n), points(oracle, "%s%d" % (tag, i), n, distinct=min(n, 16) or 1)[: 32 * n]
            rc, enc, _ = oracle.msm(s, p)
            assert rc == 0
            want.append(enc)
            sc += s
            pt += p
            offs.append(offs[-1] + n)
        assert offs[-1] <= 64 * len(sizes)                      # the average alone would choose the small path for both
        out, ok = ctx.msm_batch(sc, pt, offs)
        assert [out[32 * i: 32 * i + 32] for i in range(len(sizes))] == want, tag
        assert bits(ok, len(sizes)) == [1] * len(sizes)


@pytest.mark.parametrize("w", [4, 7, 9, 16])
def test_fixed_base_tables_equal_generic_path(ctx, oracle, w):
    """Generator terms summed out of the fixed-base window tables must give the same accept bits as the
    Pippenger path and as the oracle -- ragged rows, index lists, zero / edge scalars, bad dynamic points."""
    from zkvm_amd import PointSet
    rng = random.Random(40 + w)
    n_gen = 37
    gens = points(oracle, "tblgens", n_gen)
    ps = PointSet(ctx, gens)
    dyn_sc, dyn_pt, dyn_off = b"", b"", [0]
    st_sc, st_idx, st_off = b"", [], [0]
    gen_sc, gen_pt, gen_off = b"", b"", [0]
    B = 61
    for i in range(B):
        cnt = rng.choice([0, 1, 5, n_gen])
        idx = rng.sample(range(n_gen), cnt)
        # 0x7fffffff / 0x8000...: digits of exactly +2^(w-1) at w = 16 (stored wrapped in the int16 digit array)
        ks = [rng.choice([0, 1, L - 1, 2**252 - 1, 0x7FFFFFFF, 0x8000800080008000 << 64, rng.randrange(L), rng.randrange(L)])
              for _ in idx]
        pts = [oracle.decode(gens[32 * j: 32 * j + 32]) for j in idx]
        tot = oracle.encode(oracle.msm_points("vartime", ks, pts)) if cnt else bytes(32)
        kind = i % 6
        extra_k = (L - 1) if kind != 2 else (L - 3)          # kind 2: wrong
        extra_p = tot if kind != 4 else BAD_POINT            # kind 4: undecodable dynamic point
        n_extra = 1 if (cnt or kind in (2, 4)) else 0
        st_sc += b"".join(k.to_bytes(32, "little") for k in ks)
        st_idx += idx
        st_off.append(st_off[-1] + cnt)
        dyn_sc += extra_k.to_bytes(32, "little") * n_extra
        dyn_pt += extra_p * n_extra
        dyn_off.append(dyn_off[-1] + n_extra)
        gen_sc += b"".join(k.to_bytes(32, "little") for k in ks) + extra_k.to_bytes(32, "little") * n_extra
        gen_pt += b"".join(gens[32 * j: 32 * j + 32] for j in idx) + extra_p * n_extra
        gen_off.append(gen_off[-1] + cnt + n_extra)
    want = oracle.verify_batch(gen_sc, gen_pt, gen_off)
    plain = ctx.verify_batch_ps(ps, dyn_sc, dyn_pt, dyn_off, st_sc, st_off, static_index=st_idx)
    assert plain == want
    nbytes = ps.build_tables(w)
    assert nbytes == (255 // w + 1) * n_gen * (1 << (w - 1)) * 128      # (96-byte rows, one per 128-byte line)
    for parts in (0, 1, 3):
        ctx.set_static_parts(parts)
        assert ctx.verify_batch_ps(ps, dyn_sc, dyn_pt, dyn_off, st_sc, st_off, static_index=st_idx) == want
    ctx.set_static_parts(0)
    # static terms only (no dynamic rows at all): sum k_j G_j - k_j G_j
    k = rng.randrange(L)
    sc2 = k.to_bytes(32, "little") + (L - k).to_bytes(32, "little")
    assert ctx.verify_batch_ps(ps, b"", b"", [0, 0, 0], sc2 + sc2, [0, 2, 4], static_index=[3, 3, 5, 6]) == b"\x01"
    ps.close()


def test_msm_adversarial_bin_loads(ctx, oracle):
    """All terms in one bucket per window (equal scalars), tiny scalars (everything in window 0) and a
    half/half mix: the heavy-bin path must give the oracle's answer and must not take seconds."""
    import time
    n = 1 << 17
    pts = points(oracle, "adv", n, distinct=331)
    one_pt = pts[:32]
    k = 0x1f3e5d7c9b0a1122334455667788990011223344556677889900aabbccddeeff % L
    for name, sc in [("equal", k.to_bytes(32, "little") * n),
                     ("tiny", b"".join((i % 7 + 1).to_bytes(32, "little") for i in range(n))),
                     ("mix", k.to_bytes(32, "little") * (n // 2) + scalars("advmix", n // 2))]:
        t0 = time.perf_counter()
        got = ctx.msm(sc, pts)
        dt = time.perf_counter() - t0
        rc, want, _ = oracle.msm(sc, pts)
        assert rc == 0 and got == want, name
        assert dt < 2.0, (name, dt)
    # 2^20 equal scalars on one point: closed form (n * k) P
    n = 1 << 20
    got = ctx.msm(k.to_bytes(32, "little") * n, one_pt * n)
    assert got == ctx.msm(((n * k) % L).to_bytes(32, "little"), one_pt)


def test_msm_fat_bins_between_the_one_lane_and_the_workgroup_path(ctx, oracle):
    """Round 6: bins of 97 .. 2048 entries are summed by 16 lanes each inside the accumulation launch (kernels.hpp
    FAT_BIN / bucket_fat_role) -- in a uniform 2^20-term multiplication that is the whole top window.  Here every size class
    around both thresholds on purpose: scalars drawn from few distinct values so that a window's bins hold 90 .. 2100
    entries, mixed with uniform ones (one-lane bins) and one value shared by 5000 terms (a heavy bin); and the top window of
    uniform scalars at 2^17 terms (32 entries a bin in window 15)."""
    n = 40000
    pts = points(oracle, "fat", n, distinct=509)
    rng = random.Random(606)
    few = [rng.randrange(L) for _ in range(60)]
    sizes = [90, 96, 97, 98, 128, 255, 256, 257, 700, 2047, 2048, 2049, 2100]
    vals = []
    for i, c in enumerate(sizes):
        vals += [few[i]] * c
    vals += [few[40]] * 5000
    vals += [rng.randrange(L) for _ in range(n - len(vals))]
    rng.shuffle(vals)
    sc = b"".join(v.to_bytes(32, "little") for v in vals)
    for w in (0, 16, 13, 9):                                  # (0: the library's choice for this size)
        ctx.set_window_bits(w)
        try:
            got = ctx.msm(sc, pts)
        finally:
            ctx.set_window_bits(0)
        rc, want, _ = oracle.msm(sc, pts)
        assert rc == 0 and got == want, w
    n = 1 << 17
    sc, pts = scalars("fat top window", n), points(oracle, "fat2", n, distinct=977)
    ctx.set_window_bits(16)
    try:
        got = ctx.msm(sc, pts)
    finally:
        ctx.set_window_bits(0)
    rc, want, _ = oracle.msm(sc, pts)
    assert rc == 0 and got == want


@pytest.mark.parametrize("w", [5, 12])
def test_msm_values_over_resident_set(ctx, oracle, w):
    """zkgpu_msm_ps_batch (the prover-side primitive: Pedersen vector commitments out of the fixed-base
    tables): values equal the oracle's MSM -- whole-set rows, index lists with repeats, empty rows, edge
    scalars; a Pedersen vector commitment over the real Bulletproof generators equals the oracle's."""
    from zkvm_amd import PointSet, ZkGpuError
    rng = random.Random(70 + w)
    n_gen = 45
    gens = points(oracle, "valgens", n_gen)
    ps = PointSet(ctx, gens)
    with pytest.raises(ZkGpuError):
        ctx.msm_ps_batch(ps, (1).to_bytes(32, "little"), [0, 1])       # no tables yet
    ps.build_tables(w)
    dec = [oracle.decode(gens[32 * j: 32 * j + 32]) for j in range(n_gen)]
    sc, idx, offs, want = b"", [], [0], []
    for i in range(23):
        cnt = rng.choice([0, 1, 7, n_gen, 2 * n_gen])
        ind = [rng.randrange(n_gen) for _ in range(cnt)]
        ks = [rng.choice([0, 1, L - 1, 2**252 - 1, rng.randrange(L), rng.randrange(L)]) for _ in ind]
        sc += b"".join(k.to_bytes(32, "little") for k in ks)
        idx += ind
        offs.append(offs[-1] + cnt)
        want.append(oracle.encode(oracle.msm_points("vartime", ks, [dec[j] for j in ind])) if cnt else bytes(32))
    got = ctx.msm_ps_batch(ps, sc, offs, index=idx)
    assert [got[32 * i: 32 * i + 32] for i in range(23)] == want
    # implicit index: row m uses points 0 .. len-1
    ks = [rng.randrange(L) for _ in range(n_gen)]
    one = ctx.msm_ps_batch(ps, b"".join(k.to_bytes(32, "little") for k in ks), [0, n_gen])
    assert one == oracle.encode(oracle.msm_points("vartime", ks, dec))
    ps.close()
    # A_I-style commitment over PedersenGens + BulletproofGens(64): blinding * B_blinding + <a_L, G> + <a_R, H>
    B, Bb = ctx.pedersen_gens()
    G, H = ctx.bulletproof_gens(64)
    allp = B + Bb + G + H
    ps2 = PointSet(ctx, allp)
    ps2.build_tables(w)
    ks = [0, rng.randrange(L)] + [rng.randrange(2) for _ in range(64)] + [rng.randrange(L) for _ in range(64)]
    val = ctx.msm_ps_batch(ps2, b"".join(k.to_bytes(32, "little") for k in ks), [0, len(ks)])
    assert val == oracle.encode(oracle.msm_points("vartime", ks, [oracle.decode(allp[32 * j: 32 * j + 32]) for j in range(130)]))
    ps2.close()


def test_table_width_is_chosen_by_capacity_and_free_memory(ctx, oracle):
    """zkgpu_pointset_build_tables(.., 0): the library picks the window width -- the KNEE: the narrowest width whose addition
    count per generator term is within 19/16 of the widest feasible width's (14 bits = 10.2 GB instead of 16 bits = 34.5 GB for
    the 1026 generators of the 2-in/2-out statement on a 288 GB MI355X: 0.5 - 4 % measured), fewer for sets whose tables would
    not fit a quarter of the device -- and the verdicts are the same whatever the width (here against a set built at 9 bits)."""
    import ctypes as C
    lib = ctx.lib
    assert lib.zkgpu_choose_table_bits(ctx.h, 514) == 14
    assert lib.zkgpu_choose_table_bits(ctx.h, 1026) == 14
    w_big = lib.zkgpu_choose_table_bits(ctx.h, 200000)
    assert 4 <= w_big < 14
    assert lib.zkgpu_choose_table_bits(ctx.h, 0) == -1
    from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
    auto = BulletproofGens(ctx, 64, table_bits=-1)
    nine = BulletproofGens(ctx, 64, table_bits=9)
    try:
        assert auto.points.table_bits() == 14 and nine.points.table_bits() == 9
        assert lib.zkgpu_pointset_table_bytes(auto.points.h) == 19 * 130 * 8192 * 128
        com, proofs = oracle.cloak_prove_batch(3, 1, 1, b"auto width".ljust(32, b"\0"), threads=3)      # 1-in/1-out: 64 generators
        bad = bytearray(proofs[1]); bad[-40] ^= 1
        txs = [CloakTx(1, 1, com[128 * i: 128 * (i + 1)], bytes(bad) if i == 1 else proofs[i]) for i in range(3)]
        r = bytes(range(192))
        got = []
        for g in (auto, nine):
            v = Verifier(ctx, g)
            got.append(bits(v.verify_bitmap_gpu(txs, r), 3))
            v.close()
        assert got[0] == got[1] == [1, 0, 1]
    finally:
        auto.close()
        nine.close()
