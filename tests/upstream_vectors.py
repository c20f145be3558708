"""Loader and checkers for vectors in the schema of tests/golden/upstream/README.md.

Two back ends check the same vector: `check_oracle` (the CPU restatement, oracle/) and `check_hip` (the product through its
C ABI).  tests/test_upstream_vectors.py runs them over tests/golden/upstream/*.json -- skipped while that directory is
empty -- and over the self-generated example, which pins nothing and only proves that the harness works."""
import glob
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
UPSTREAM_DIR = os.path.join(HERE, "golden", "upstream")
EXAMPLE = os.path.join(HERE, "golden", "upstream_example", "self_generated.json")
L = 2**252 + 27742317777372353535851937790883648493
DEFAULT_R = hashlib.shake_256(b"zkvm_amd upstream-vector harness: verifier weight").digest(64)
KINDS = ("generators", "transcript", "cloak", "tx")
TX_EXPECT = {"accept": 0, "reject": 1, "outside-subset": 2}


def h(x):
    return bytes.fromhex(x)


def load(path):
    """-> [(file, source, vector)] of one file or of every *.json of a directory; validates the schema"""
    files = sorted(glob.glob(os.path.join(path, "*.json"))) if os.path.isdir(path) else [path]
    out = []
    for f in files:
        doc = json.load(open(f))
        assert isinstance(doc.get("source"), str) and doc["source"], (f, "a file names its source")
        assert isinstance(doc.get("vectors"), list) and doc["vectors"], (f, "no vectors")
        for v in doc["vectors"]:
            assert v.get("kind") in KINDS and isinstance(v.get("name"), str), (f, v.get("name"), v.get("kind"))
            if v["kind"] == "cloak":
                assert len(h(v["commitments"])) == 64 * (v["n_in"] + v["n_out"]) and v["expect"] in ("accept", "reject"), (f, v["name"])
            if v["kind"] == "tx":
                assert v["expect"] in TX_EXPECT, (f, v["name"])
            out.append((os.path.basename(f), doc["source"], v))
    return out


def upstream_present():
    return bool(glob.glob(os.path.join(UPSTREAM_DIR, "*.json")))


def _challenge_list(v):
    c = v.get("challenges")
    if not c:
        return None
    return [int.from_bytes(h(x), "little") for x in c.get("phase2", [])] + \
           [int.from_bytes(h(c[k]), "little") for k in ("y", "z", "u", "x", "w")] + [int.from_bytes(h(x), "little") for x in c.get("ipp", [])]


# ---- the oracle -------------------------------------------------------------------------------------------------------
def check_oracle(oracle, v):
    k = v["kind"]
    if k == "generators":
        B, Bb = oracle.pedersen_gens()
        assert Bb == h(v["B_blinding"]), "PedersenGens::default().B_blinding"
        for which in "GH":
            want = [h(x) for x in v.get(which, [])]
            assert oracle.bulletproof_gens(len(want), which) == want, "BulletproofGens %s" % which
    elif k == "transcript":
        t = oracle.MerlinTranscript(v["label"].encode())
        for op in v["ops"]:
            if op[0] == "message":
                t.append_message(op[1].encode(), h(op[2]))
            elif op[0] == "u64":
                t.append_u64(op[1].encode(), op[2])
            elif op[0] == "challenge_bytes":
                assert t.challenge_bytes(op[1].encode(), op[2]) == h(op[3]), (v["name"], op[1])
            elif op[0] == "challenge_scalar":
                assert t.challenge_scalar(op[1].encode()) == int.from_bytes(h(op[2]), "little"), (v["name"], op[1])
            else:
                raise AssertionError("unknown transcript op %r" % op[0])
    elif k == "cloak":
        com, proof, r = h(v["commitments"]), h(v["proof"]), h(v.get("verifier_r", DEFAULT_R.hex()))
        want = _challenge_list(v)
        if want is not None:
            got = oracle.cloak_verify_challenges(com, v["n_in"], v["n_out"], proof, r)
            assert got is not None, (v["name"], "the oracle calls the proof malformed")
            for i, (a, b) in enumerate(zip(got, want)):
                assert a == b, (v["name"], "challenge %d of the transcript differs (order: phase2.., y z u x w, ipp..)" % i)
            assert len(got) == len(want), v["name"]
        assert oracle.cloak_verify(com, v["n_in"], v["n_out"], proof, r) == (v["expect"] == "accept"), v["name"]
    elif k == "tx":
        tx = h(v["tx"])
        if "txid" in v:
            st, txid, _, _ = oracle.tx_id(tx)
            assert st == 0 and txid == h(v["txid"]), (v["name"], "transaction ID")
        assert oracle.tx_verify(tx, h(v.get("verifier_r", DEFAULT_R.hex()))) == TX_EXPECT[v["expect"]], v["name"]


# ---- the product, through its C ABI -----------------------------------------------------------------------------------
class HipChecker:
    """one context, one table set per generator capacity, one block verifier for transactions"""

    def __init__(self, ctx):
        self.ctx, self.gens, self.verifiers, self.bv = ctx, {}, {}, None

    def _gens(self, cap):
        from zkvm_amd.verifier import BulletproofGens, Verifier
        if cap not in self.gens:
            self.gens[cap] = BulletproofGens(self.ctx, cap, table_bits=8)
            self.verifiers[cap] = Verifier(self.ctx, self.gens[cap])
        return self.gens[cap], self.verifiers[cap]

    def close(self):
        if self.bv is not None:
            self.bv.close()
        for v in self.verifiers.values():
            v.close()
        for g in self.gens.values():
            g.close()

    def check(self, v):
        k = v["kind"]
        ctx = self.ctx
        if k == "generators":
            B, Bb = ctx.pedersen_gens()
            assert Bb == h(v["B_blinding"]), "PedersenGens::default().B_blinding"
            n = max(len(v.get("G", [])), len(v.get("H", [])), 1)
            G, H = ctx.bulletproof_gens(n)
            for which, got in (("G", G), ("H", H)):
                want = b"".join(h(x) for x in v.get(which, []))
                assert got[: len(want)] == want, "BulletproofGens %s" % which
        elif k == "transcript":
            return "host-only"          # (the product's Merlin is exercised through cloak / tx vectors and tests/test_host_logic.py)
        elif k == "cloak":
            cap = v.get("bp_gens_capacity", 256)
            _, ver = self._gens(cap)
            com, proof, r = h(v["commitments"]), h(v["proof"]), h(v.get("verifier_r", DEFAULT_R.hex()))
            from gpu_util import bits
            got = bits(ver.verify_packed_gpu(v["n_in"], v["n_out"], 1, com, proof, len(proof), r), 1)[0]
            want = _challenge_list(v)
            if want is not None and got == 1:      # (the device's challenge slots of the batch just run: tests/test_gpu_block.py reads them the same way)
                lay = ver.plan_layout(v["n_in"], v["n_out"])
                ch = ctx.debug_read("challenges", lay["slots"] * 32)
                inv = pow(pow(2, 260, L), -1, L)
                slot = lambda j: int.from_bytes(ch[32 * j: 32 * j + 32], "little") * inv % L          # noqa: E731
                n2, kk = lay["n_chal2"], lay["k"]
                dev = [slot(14 + j) for j in range(n2)] + [slot(j) for j in range(5)] + [slot(14 + n2 + j) for j in range(kk)]
                for i, (a, b) in enumerate(zip(dev, want)):
                    assert a == b, (v["name"], "device challenge %d differs" % i)
            assert got == (1 if v["expect"] == "accept" else 0), v["name"]
        elif k == "tx":
            from zkvm_amd.verifier import BlockVerifier
            if self.bv is None:
                g, _ = self._gens(256)
                self.bv = BlockVerifier(ctx, g, batches_in_flight=2)
                self.bv.set_tx_format(BlockVerifier.TXFORMAT_RECOLLECTED_V1)
            bm, status = self.bv.verify_txs([h(v["tx"])])
            assert status[0] == TX_EXPECT[v["expect"]] and (bm[0] & 1) == (1 if v["expect"] == "accept" else 0), v["name"]
        return "checked"
