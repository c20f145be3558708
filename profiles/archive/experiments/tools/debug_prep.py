"""GPU debug: device-side challenges of the first golden proofs vs the host tape replay."""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")   # the profile / mode hooks (include/zkgpu_hooks.h) are not exports
import ctypes as C, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zkvm_amd import Context
from zkvm_amd.verifier import BulletproofGens, Verifier
L = 2**252 + 27742317777372353535851937790883648493
RINV = pow(2**256, -1, L)
raw = open(os.path.join(ROOT, "tests/golden/cloak_2x2_proofs.bin"), "rb").read()
count, n_in, n_out, plen = struct.unpack("<IIII", raw[8:24])
w = 64 * (n_in + n_out); rec = w + plen
txs = [(raw[24 + rec * i: 24 + rec * i + w], raw[24 + rec * i + w: 24 + rec * (i + 1)]) for i in range(4)]
host = C.CDLL(os.path.join(ROOT, "zkvm_amd/lib/libzkhost.so"))
ctx = Context(0)
ctx.lib.zkgpu_debug_read.restype = C.c_longlong
ctx.lib.zkgpu_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
ctx.set_group_size(1)
gens = BulletproofGens(ctx, 256, table_bits=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
v = Verifier(ctx, gens)
r = bytes(range(64)) * len(txs)
bm = v.verify_packed_gpu(2, 2, len(txs), b"".join(t[0] for t in txs), b"".join(t[1] for t in txs), plen, r)
print("bitmap", bm.hex())
n_ch_ext = 14 + 8 + 16 + 15 + 48
buf = C.create_string_buffer(len(txs) * n_ch_ext * 32)
print("read", ctx.lib.zkgpu_debug_read(ctx.h, b"challenges", buf, len(buf)))
for i, (com, proof) in enumerate(txs[:2]):
    a, b = C.create_string_buffer(32 * 64), C.create_string_buffer(32 * 64)
    n = host.zkhost_tape_challenges(2, 2, com, proof, C.c_size_t(len(proof)), a, b, C.c_size_t(64))
    want = [int.from_bytes(b.raw[32 * j: 32 * j + 32], "little") for j in range(n)]
    slots = [0, 1, 2, 3, 4] + [14 + j for j in range(8)] + [22 + j for j in range(8)]
    got = [int.from_bytes(buf.raw[(i * n_ch_ext + s) * 32: (i * n_ch_ext + s) * 32 + 32], "little") * RINV % L for s in slots]
    print("tx", i, "challenge match:", [int(x == y) for x, y in zip(got, want)])
    U = 1
    for j in range(8):
        U = U * want[13 + j] ** 2 % L
    gU = int.from_bytes(buf.raw[(i * n_ch_ext + 6) * 32: (i * n_ch_ext + 6) * 32 + 32], "little") * RINV % L
    print("   U ok:", gU == U)

ns, nd = 514, 35
sbuf = C.create_string_buffer(len(txs) * ns * 32); dbuf = C.create_string_buffer(len(txs) * nd * 32)
ctx.lib.zkgpu_debug_read(ctx.h, b"static_scalars", sbuf, len(sbuf)); ctx.lib.zkgpu_debug_read(ctx.h, b"dyn_scalars", dbuf, len(dbuf))
for i, (com, proof) in enumerate(txs[:1]):
    a, b = C.create_string_buffer(32 * 64), C.create_string_buffer(32 * 64)
    n = host.zkhost_tape_challenges(2, 2, com, proof, C.c_size_t(len(proof)), a, b, C.c_size_t(64))
    ch = [int.from_bytes(b.raw[32 * j: 32 * j + 32], "little") for j in range(n)]
    y = ch[0]; U = 1
    for j in range(8):
        U = U * ch[13 + j] ** 2 % L
    cp = pow(y, 255, L) * U % L
    hd, hp = C.create_string_buffer(32 * 64), C.create_string_buffer(32 * 64)
    hs, hi = C.create_string_buffer(32 * 600), (C.c_uint32 * 600)()
    n_dyn, n_st, pn = C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = host.zkhost_cloak_prepare(com, C.c_size_t(2), C.c_size_t(2), proof, C.c_size_t(len(proof)), r[64 * i: 64 * i + 64],
                                   C.c_size_t(256), hd, hp, C.byref(n_dyn), hs, hi, C.byref(n_st), C.byref(pn))
    print("host prepare rc", rc, n_dyn.value, n_st.value, pn.value)
    bad_s = [j for j in range(ns) if int.from_bytes(sbuf.raw[(i * ns + j) * 32:(i * ns + j) * 32 + 32], "little")
             != int.from_bytes(hs.raw[32 * j: 32 * j + 32], "little") * cp % L]
    bad_d = [j for j in range(nd) if int.from_bytes(dbuf.raw[(i * nd + j) * 32:(i * nd + j) * 32 + 32], "little")
             != int.from_bytes(hd.raw[32 * j: 32 * j + 32], "little") * cp % L]
    print("static mismatches:", len(bad_s), bad_s[:20], "...", bad_s[-5:])
    print("dyn mismatches:", len(bad_d), bad_d)

# pinpoint: g_i for i >= n is -(a s_i) c' u
com, proof = txs[0]
f = proof[1:]
k = 8
a_ = int.from_bytes(f[32 * (14 + 2 * k): 32 * (14 + 2 * k) + 32], "little")
b_ = int.from_bytes(f[32 * (15 + 2 * k): 32 * (15 + 2 * k) + 32], "little")
us = ch[13:21]; u = ch[2]
def s_of(i):
    v = 1
    for j in range(k):
        bit = (i >> (k - 1 - j)) & 1
        v = v * (us[j] if bit else pow(us[j], -1, L)) % L
    return v
for i in (200, 255, 140):
    dev = int.from_bytes(sbuf.raw[(2 + i) * 32:(2 + i) * 32 + 32], "little")
    hostv = int.from_bytes(hs.raw[32 * (2 + i): 32 * (2 + i) + 32], "little")
    cands = {"-a s c u": (-a_ * s_of(i)) * cp * u % L, "-a s c": (-a_ * s_of(i)) * cp % L, "host*cp": hostv * cp % L,
             "-a s u (host?)": (-a_ * s_of(i)) * u % L}
    print(i, {k2: int(v2 == dev) for k2, v2 in cands.items()}, "host==-a s u:", hostv == (-a_ * s_of(i)) * u % L)
    # ratio dev / (host*cp)
    ratio = dev * pow(hostv * cp % L, -1, L) % L
    print("   ratio dev/(host cp) =", hex(ratio)[:20], " == 1/U?", ratio == pow(U, -1, L), " == U?", ratio == U, "==1/y^255?", ratio == pow(pow(y,255,L),-1,L))
R = 2**256 % L
dev = int.from_bytes(sbuf.raw[(2 + 255) * 32:(2 + 255) * 32 + 32], "little")
hostv = int.from_bytes(hs.raw[32 * (2 + 255): 32 * (2 + 255) + 32], "little")
ratio = dev * pow(hostv * cp % L, -1, L) % L
Y = pow(y, 255, L)
for name, val in {"R": R, "1/R": pow(R, -1, L), "R^2": R * R % L, "1/Y": pow(Y, -1, L), "Y": Y, "1/(U)": pow(U, -1, L),
                  "1/cp": pow(cp, -1, L), "y": y, "1/y": pow(y, -1, L), "U/Y": U * pow(Y, -1, L) % L,
                  "prod 1/u": pow(U, -1, L), "1/(YU) ": pow(Y * U % L, -1, L)}.items():
    print(name, ratio == val)
# is dev == -a * u * X for simple X?
base = dev * pow((-a_ * u) % L, -1, L) % L
print("dev/(-a u) ==", {"Y*U": base == Y * U % L, "Y": base == Y, "U": base == U, "1": base == 1, "Y*R": base == Y * R % L,
                        "U*U*Y/..": base == U * U % L * Y % L, "Y*prod_u": False})
