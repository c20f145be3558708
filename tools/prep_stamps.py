"""Where k_prepare spends its time (first half): a library built with -DZK_PREP_STAMPS
(tools/build_variant.sh stamps -DZK_PREP_STAMPS; run with ZKGPU_LIB=build/ab/stamps/libzkgpu.so) notes s_memtime at the end
of every section, thread 0 and thread 255 of workgroups 4096 .. 4351 of one 10 240-transaction launch.
usage: prep_stamps.py [serial=1]   (serial: every kernel alone on the chip, as `bench.py --solo` times them)"""
import os as _os; _os.environ.setdefault("ZKGPU_TEST_HOOKS", "1")
import ctypes as C
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "18")
from gpu_util import benched_randomness, benched_step
from zkvm_amd import Context
from zkvm_amd.verifier import BlockVerifier, BulletproofGens
serial = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = 10240
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=-1)
bv = BlockVerifier(ctx, gens, batches_in_flight=5)
bv.set_merge(batch)
for i in range(bv.lanes()):
    bv.lane(i).set_serial(serial)
txs, expected = benched_step(1024, 0, 64, 0)
k = batch // 1024
com, proofs = b"".join(t[2] for t in txs) * k, b"".join(t[3] for t in txs) * k
r = benched_randomness(0, 0, 1024) * k
d = [ctx.to_device(x) for x in (com, proofs, r)]
SECTIONS = ["slots -> limb form", "z powers (doubling)", "plan replay (products, target sums)", "y / s tables (doubling)", "dsum, c'",
            "proof-point scalars (thread 255) / - (thread 0)", "generator scalars"]
for rep in range(3):
    bv.wait(bv.submit_dev(2, 2, batch, d[0], d[1], len(txs[0][3]), d[2]))
import struct
raw = ctx.debug_read("prep_stamps", 256 * 2 * 16 * 8)        # (a __device__ array of the library: any context reads it)
assert len(raw) == 256 * 2 * 16 * 8, "this library was not built with -DZK_PREP_STAMPS"
buf = struct.unpack("<%dQ" % (256 * 2 * 16), raw)
for who, name in ((0, "thread 0 (wavefront 0)"), (1, "thread 255 (the last wavefront)")):
    rows = []
    for wg in range(256):
        s = [buf[(wg * 2 + who) * 16 + i] for i in range(8)]
        if all(s) and s[7] > s[0]:
            rows.append([s[i + 1] - s[i] for i in range(7)] + [s[7] - s[0]])
    print("%s: %d workgroups, s_memtime ticks (median; share of the workgroup's life)" % (name, len(rows)))
    tot = statistics.median(x[7] for x in rows)
    for i, sec in enumerate(SECTIONS):
        m = statistics.median(x[i] for x in rows)
        print("  %-52s %9.0f  %5.1f %%" % (sec, m, 100.0 * m / tot))
    print("  %-52s %9.0f" % ("whole workgroup", tot))
    inner = []
    for wg in range(256):
        s = [buf[(wg * 2 + who) * 16 + i] for i in range(12)]
        if all(s[i] for i in (1, 2, 3, 8, 9, 11)):
            inner.append((s[11] - s[1], s[2] - s[11], s[8] - s[2], s[9] - s[8], s[3] - s[9]))
    if inner:
        for i, sec in enumerate(["small tables (levels)", "z^(q+1): one product per entry", "plan replay: products of the (last) pass", "plan replay: sums of the light targets",
                                 "plan replay: heavy targets"]):
            print("    %-50s %9.0f" % (sec, statistics.median(x[i] for x in inner)))
bv.close(); gens.close(); ctx.close()
