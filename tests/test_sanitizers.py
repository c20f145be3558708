"""CPU tier of SURVEY.md sec 5 (race / memory-error detection; GPU AddressSanitizer is not available on the
pool): the product's host logic (zkvm_amd/csrc/hostlib.cpp -> the same headers libzkgpu.so compiles) and the
oracle, built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through a proof verification, a
malformed-proof sweep and the prover in a child process (the sanitizer runtime must be loaded first)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_build")

DRIVER = textwrap.dedent(r"""
    import ctypes as C, os, sys, struct
    root, host_so, oracle_so = sys.argv[1:4]
    host, orc = C.CDLL(host_so), C.CDLL(oracle_so)
    raw = open(os.path.join(root, "tests", "golden", "cloak_mixed.bin"), "rb").read()
    pos, fix = 12, {}
    for _ in range(struct.unpack("<I", raw[8:12])[0]):
        count, n_in, n_out, plen = struct.unpack("<IIII", raw[pos:pos + 16]); pos += 16
        w = 64 * (n_in + n_out); recs = []
        for _ in range(count):
            recs.append((raw[pos:pos + w], raw[pos + w:pos + w + plen])); pos += w + plen
        fix[(n_in, n_out)] = recs
    r = bytes(range(64))
    def host_prepare(com, n_in, n_out, proof, cap=512):
        ds, dp = C.create_string_buffer(32 * 128), C.create_string_buffer(32 * 128)
        ss, si = C.create_string_buffer(32 * (2 + 2 * cap)), (C.c_uint32 * (2 + 2 * cap))()
        nd, ns, pn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        return host.zkhost_cloak_prepare(com, C.c_size_t(n_in), C.c_size_t(n_out), proof, C.c_size_t(len(proof)), r,
                                         C.c_size_t(cap), ds, dp, C.byref(nd), ss, si, C.byref(ns), C.byref(pn))
    for (n_in, n_out), recs in sorted(fix.items()):
        com, proof = recs[0]
        assert host_prepare(com, n_in, n_out, proof) == 0
        assert orc.zko_cloak_verify(com, C.c_size_t(n_in), C.c_size_t(n_out), proof, C.c_size_t(len(proof)), r) == 1
        # malformed inputs: every truncation to a multiple of 32 (+1), empty, version byte, oversized
        for cut in list(range(0, len(proof), 97)) + [1, 33, len(proof) - 32, len(proof) - 1]:
            bad = proof[:cut]
            assert host_prepare(com, n_in, n_out, bad) != 0
            assert orc.zko_cloak_verify(com, C.c_size_t(n_in), C.c_size_t(n_out), bad, C.c_size_t(len(bad)), r) == 0
        assert host_prepare(com, n_in, n_out, proof + bytes(64)) != 0
        assert host_prepare(com, n_in, n_out, b"\x07" + proof[1:]) != 0
        assert host_prepare(com, n_in, n_out, proof, cap=8) != 0
    # the oracle prover (and the product prover's host half through zkhost, if exported)
    com = C.create_string_buffer(64 * 2); pr = C.create_string_buffer(4096); plen = C.c_size_t(0)
    q = (C.c_uint64 * 2)(5, 5); fl = bytes(range(32)) * 2
    fl = bytes([1] + [0] * 31) * 2
    assert orc.zko_cloak_prove(q, fl, C.c_size_t(1), C.c_size_t(1), bytes(32), com, pr, C.c_size_t(4096), C.byref(plen), None) == 0
    assert orc.zko_cloak_verify(com.raw, C.c_size_t(1), C.c_size_t(1), pr.raw[:plen.value], plen, r) == 1
    out = C.create_string_buffer(32)
    for op in range(6):
        assert host.zkhost_scalar_op(op, bytes([255]) * 64, bytes(range(64)), out) == 0
    # the device prover's phase functions (prover_dev.hpp) run on the host: same proof as the oracle's
    class Ge(C.Structure):
        _fields_ = [("v", C.c_uint64 * 20)]
    def enc(g):
        o = C.create_string_buffer(32); orc.ristretto_encode(o, C.byref(g)); return o.raw
    b_, bb_ = Ge(), Ge()
    orc.pedersen_gens(C.byref(b_), C.byref(bb_))
    gs, hs = (Ge * 64)(), (Ge * 64)()
    orc.bulletproof_gens_chain(gs, C.c_size_t(64), C.c_char(b"G"), C.c_uint32(0))
    orc.bulletproof_gens_chain(hs, C.c_size_t(64), C.c_char(b"H"), C.c_uint32(0))
    gens = enc(b_) + enc(bb_) + b"".join(enc(gs[i]) for i in range(64)) + b"".join(enc(hs[i]) for i in range(64))
    com2 = C.create_string_buffer(64 * 2); pr2 = C.create_string_buffer(4096); plen2 = C.c_size_t(0)
    assert host.zkhost_prove_dev_cloak(1, 1, q, fl, bytes(32), gens, C.c_size_t(64), com2, pr2, C.c_size_t(4096), C.byref(plen2)) == 0
    assert com2.raw == com.raw and pr2.raw[:plen2.value] == pr.raw[:plen.value]
    # serialized transactions: the oracle builds one, both sides read it, then every truncation and a byte flip every 13 bytes
    orc.zko_tx_build_payment.restype = C.c_size_t
    txb = C.create_string_buffer(65536)
    q4 = (C.c_uint64 * 4)(5, 9, 4, 10)
    n = orc.zko_tx_build_payment(C.c_size_t(2), C.c_size_t(2), q4, bytes([1] + [0] * 31) * 4, bytes(range(32)), C.c_uint64(1), C.c_uint64(9), txb, C.c_size_t(65536))
    tx = txb.raw[:n]
    assert n > 1000 and orc.zko_tx_verify(tx, C.c_size_t(n), r) == 0
    def host_tx(t):
        txid = C.create_string_buffer(32); a, b = C.c_uint32(0), C.c_uint32(0); cm = C.create_string_buffer(64 * 128)
        ss, sp = C.create_string_buffer(32 * 80), C.create_string_buffer(32 * 80)
        ns, po, pl = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        return host.zkhost_tx_prepare(t, C.c_size_t(len(t)), txid, C.byref(a), C.byref(b), cm, C.c_size_t(64 * 128), ss, sp, C.c_size_t(80),
                                      C.byref(ns), C.byref(po), C.byref(pl)), txid.raw
    tid = C.create_string_buffer(32)
    assert orc.zko_tx_id(tx, C.c_size_t(n), tid, None, None) == 0
    assert host_tx(tx) == (0, tid.raw)
    for cut in list(range(0, n, 41)) + [n - 1]:
        assert host_tx(tx[:cut])[0] != 0 and orc.zko_tx_verify(tx[:cut], C.c_size_t(cut), r) != 0
    for at in range(24, n - 1100, 13):                                     # header, program and signature bytes
        bad = bytearray(tx); bad[at] ^= 0x81
        bad = bytes(bad)
        assert orc.zko_tx_verify(bad, C.c_size_t(n), r) != 0
        host_tx(bad)
    print("sanitized run ok")
""")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(600)
def test_host_logic_and_oracle_under_asan_ubsan(tmp_path):
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_build/liboracle_asan.so"], check=True, capture_output=True)
    host_so = os.path.join(OUT, "libzkhost_asan.so")
    src = os.path.join(ROOT, "zkvm_amd", "csrc", "hostlib.cpp")
    deps = [os.path.join(ROOT, "zkvm_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "zkvm_amd", "csrc"))]
    if not os.path.exists(host_so) or any(os.path.getmtime(d) > os.path.getmtime(host_so) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-o", host_so, src], check=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", DRIVER, ROOT, host_so, os.path.join(OUT, "liboracle_asan.so")], env=env,
                       capture_output=True, text=True, timeout=500)
    assert p.returncode == 0 and "sanitized run ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


@pytest.mark.timeout(600)
def test_host_worker_pool_under_tsan(tmp_path):
    """host_pool.hpp under ThreadSanitizer: its selftest (hostlib.cpp, zkhost_pool_selftest: four callers at once, every size
    and thread count) as a standalone program -- the race-detection tier for the one piece of the product that shares
    state between host threads without going through a context's mutex."""
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(tsan) and os.path.exists(tsan)):
        pytest.skip("gcc has no libtsan here")
    main = tmp_path / "main.cpp"
    main.write_text('extern "C" unsigned long long zkhost_pool_selftest(unsigned, unsigned);\n'
                    'int main() { return zkhost_pool_selftest(2, 4) ? 1 : 0; }\n')
    exe = tmp_path / "pool_tsan"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", str(main),
                    os.path.join(ROOT, "zkvm_amd", "csrc", "hostlib.cpp"), "-o", str(exe)], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]


TXCALL_MAIN = r"""
// drives zkhost_txcall_selftest (hostlib.cpp: TxCall on the stand-in device) from a file the test wrote:
//   u64 batch | u64 blob bytes | offsets[batch + 1] u64 | proof_ok[batch] | expected[batch] | blob
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
extern "C" int zkhost_txcall_selftest(size_t, const uint8_t*, const uint64_t*, const uint8_t*, int, size_t, uint32_t, int, uint8_t*, uint8_t*,
                                      size_t*, size_t*, size_t*);
extern "C" int zkhost_txcall_pair_selftest(size_t, size_t, const uint8_t*, const uint64_t*, const uint8_t*, int, size_t, uint32_t, uint8_t*, uint8_t*, size_t*);
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t batch = 0, blob_len = 0;
  if (fread(&batch, 8, 1, f) != 1 || fread(&blob_len, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> offs(batch + 1);
  std::vector<uint8_t> proof_ok(batch), expected(batch), blob(blob_len);
  if (fread(offs.data(), 8, batch + 1, f) != batch + 1 || fread(proof_ok.data(), 1, batch, f) != batch ||
      fread(expected.data(), 1, batch, f) != batch || fread(blob.data(), 1, blob_len, f) != blob_len) return 2;
  fclose(f);
  std::vector<uint8_t> bm((batch + 7) / 8 + 1), st(batch);
  size_t nc = 0, ns = 0, leaked = 0;
  const size_t chunks[] = {0, 8, 16, 24};
  for (int round = 0; round < 4; ++round) {
    int rc = zkhost_txcall_selftest(batch, blob.data(), offs.data(), proof_ok.data(), 4, chunks[round], 40u + round, -1, bm.data(), st.data(), &nc, &ns, &leaked);
    if (rc != 0 || leaked != 0) { printf("round %d: rc %d leaked %zu\n", round, rc, leaked); return 1; }
    for (uint64_t i = 0; i < batch; ++i)
      if (((bm[i / 8] >> (i % 8)) & 1) != expected[i]) { printf("round %d: bit %llu differs\n", round, (unsigned long long)i); return 1; }
  }
  // two calls stepped by one thread, one stage slot each (the engine's way of keeping two rounds in flight)
  for (int round = 0; round < 3; ++round) {
    const size_t splits[] = {32, 8, 48};
    int rc = zkhost_txcall_pair_selftest(batch, splits[round], blob.data(), offs.data(), proof_ok.data(), 4, chunks[round + 1], 70u + round, bm.data(), st.data(), &leaked);
    if (rc != 0 || leaked != 0) { printf("pair %d: rc %d leaked %zu\n", round, rc, leaked); return 1; }
    for (uint64_t i = 0; i < batch; ++i)
      if (((bm[i / 8] >> (i % 8)) & 1) != expected[i]) { printf("pair %d: bit %llu differs\n", round, (unsigned long long)i); return 1; }
  }
  // faults: every third device operation of a many-chunk call
  for (int k = 0; k < 40; k += 3) {
    int rc = zkhost_txcall_selftest(batch, blob.data(), offs.data(), proof_ok.data(), 4, 16, 90u + k, k, bm.data(), st.data(), &nc, &ns, &leaked);
    if (leaked != 0) { printf("fault %d: leaked %zu\n", k, leaked); return 1; }
    if (rc != 0) for (uint64_t i = 0; i < (batch + 7) / 8; ++i) if (bm[i]) { printf("fault %d: a bit survived\n", k); return 1; }
  }
  printf("txcall ok\n");
  return 0;
}
"""


def _txcall_input(path, count=72):
    import struct
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import built_transactions
    txs, expected = built_transactions(count, call=11, bad_every=8)
    clean, _ = built_transactions(count, call=11, bad_every=0)
    proof_ok = []
    for a, b in zip(txs, clean):
        plen = struct.unpack("<I", b[24:28])[0]
        proof_ok.append(0 if a[28 + plen + 64:] != b[28 + plen + 64:] else 1)
    blob = b"".join(txs)
    offs = [0]
    for t in txs:
        offs.append(offs[-1] + len(t))
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", count, len(blob)) + struct.pack("<%dQ" % (count + 1), *offs) + bytes(proof_ok) + bytes(expected) + blob)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_transaction_call_scheduling_under_tsan_and_asan(tmp_path, sanitizer):
    """csrc/tx_call.hpp -- the two threads of a zkgpu_tx_verify_batch call, their flags, the ring of staging areas, the stage
    slots, the error paths -- under ThreadSanitizer and under AddressSanitizer + UBSan, on the stand-in device whose stages
    complete on threads of their own after random delays ("no CPU-side test of its scheduling"): one-chunk and
    many-chunk calls with the verdicts checked, then device faults injected along a many-chunk call."""
    lib = "libtsan.so" if sanitizer == "thread" else "libasan.so"
    have = subprocess.run(["gcc", "-print-file-name=" + lib], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(have) and os.path.exists(have)):
        pytest.skip("gcc has no %s here" % lib)
    main = tmp_path / "main.cpp"
    main.write_text(TXCALL_MAIN)
    exe = tmp_path / "txcall_san"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-pthread", str(main),
                    os.path.join(ROOT, "zkvm_amd", "csrc", "hostlib.cpp"), "-o", str(exe)], check=True)
    data = tmp_path / "txs.bin"
    _txcall_input(str(data))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([str(exe), str(data)], capture_output=True, text=True, timeout=800, env=env)
    assert p.returncode == 0 and "txcall ok" in p.stdout and "Sanitizer" not in p.stderr, (p.stdout[-1500:], p.stderr[-4000:])
