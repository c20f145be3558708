#!/usr/bin/env python3
"""The manifest of everything in this repository that is a RECOLLECTION of upstream (slingshot `zkvm`, `spacesuit`,
dalek `bulletproofs` / `merlin`) and that no vector or source held here pins: every Merlin label and its place in the
R1CS transcript, the generator-chain labels, the R1CSProof wire format, the cloak gadget's variable order, the
transaction format's opcodes and labels (SURVEY.md sec 8 (c) "Not pinned by anything here", (f-1)).

    python tools/unpinned_manifest.py            # rewrite tests/golden/unpinned_manifest.json from the sources
    python tools/unpinned_manifest.py --check    # exit 1 if the committed file differs from what the sources give

The CATALOGUE below is written by hand: what each constant is, which upstream item it restates (a recollected name, not
a file:line -- /root/reference holds no source) and what would pin it.  The LOCATIONS (file:line in the product AND in
the oracle) are found by scanning the sources, so they cannot go stale silently: tests/test_unpinned_manifest.py runs the
same scan and fails when (1) a label literal of a protocol source file on either side is neither in the catalogue nor in
its list of literals that are this repository's own, (2) a catalogued constant is no longer found on one side, (3) the
committed JSON is not what the scan gives, (4) the label ORDER of the R1CS verifier transcript differs between the
manifest, the product's host verifier, the product's device tape and the oracle.

An integrator who gets the real sources diffs this ONE file against them (INTEGRATION.md "Bringing real vectors").
"""
from __future__ import annotations

import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "unpinned_manifest.json")

# protocol sources: EVERY short string literal in their code (comments, error texts and includes apart) must be classified
PRODUCT_STRICT = ["zkvm_amd/csrc/r1cs_verifier.hpp", "zkvm_amd/csrc/transcript_tape.hpp", "zkvm_amd/csrc/merlin.hpp",
                  "zkvm_amd/csrc/cloak_plan.hpp", "zkvm_amd/csrc/r1cs_prover.hpp", "zkvm_amd/csrc/prover_plan.hpp",
                  "zkvm_amd/csrc/prover_dev.hpp", "zkvm_amd/csrc/zkvm_tx.hpp"]
ORACLE_STRICT = ["oracle/r1cs.c", "oracle/merlin.c", "oracle/cloak.c", "oracle/zkvm_tx.c"]
# searched for catalogued values only (large files full of other text)
PRODUCT_EXTRA = ["zkvm_amd/csrc/zkgpu.hip", "zkvm_amd/csrc/prep_kernels.hpp", "zkvm_amd/csrc/zkvm_tx_build.hpp", "zkvm_amd/csrc/curve.hpp",
                 "zkvm_amd/csrc/session.hpp"]
ORACLE_EXTRA = ["oracle/pyref.py", "oracle/gadgets.c", "oracle/ristretto.c"]

# lines that hold text for people, not protocol bytes
NOT_CODE = re.compile(r"#\s*include|last_error|throw\s|fprintf|printf\(|getenv|\bfail\(|snprintf|\.what\(\)|strcmp\(|assert|return\s+\"|perror|TX_INVALID|TX_UNSUPPORTED")

PINNED_MERLIN = "the Merlin 'test protocol' known answer (tests/test_oracle_transcript.py) pins the STROBE/Merlin framing; the label itself is a recollection"


def L(id, kind, values, role, upstream, pinned_by="nothing held here", sides="both"):
    return {"id": id, "kind": kind, "values": list(values), "role": role, "upstream": upstream, "pinned_by": pinned_by, "sides": sides}


CATALOGUE = [
    # ---- Merlin / STROBE framing (spec'd publicly; pinned by the Merlin KAT) -------------------------------------------
    L("strobe.version", "protocol-constant", ["STROBEv1.0.2"], "STROBE-128 initial state string", "strobe-rs / merlin::strobe::Strobe128::new",
      "STROBE v1.0.2 specification + the Merlin known answer"),
    L("merlin.protocol", "protocol-constant", ["Merlin v1.0"], "Merlin protocol label mixed in by Transcript::new", "merlin::Transcript::new",
      "Merlin known answer (tests/test_oracle_transcript.py)"),
    L("merlin.domsep", "merlin-label", ["dom-sep"], "label of every domain-separator message (Transcript::new, r1cs, ipp, signature)", "merlin::Transcript::new; bulletproofs transcript.rs",
      PINNED_MERLIN),
    L("merlin.rng", "merlin-label", ["rng"], "meta-AD label with which TranscriptRngBuilder::finalize keys the RNG", "merlin::TranscriptRngBuilder::finalize"),
    # ---- R1CS proof transcript (bulletproofs, feature yoloproofs) ---------------------------------------------------------
    L("r1cs.domsep", "domain-separator", ["r1cs v1"], "dom-sep message when a ConstraintSystem is created", "bulletproofs::transcript::TranscriptProtocol::r1cs_domain_sep"),
    L("r1cs.1phase", "domain-separator", ["r1cs-1phase"], "dom-sep after A_I1 A_O1 S1 when there are NO randomized constraints", "TranscriptProtocol::r1cs_1phase_domain_sep"),
    L("r1cs.2phase", "domain-separator", ["r1cs-2phase"], "dom-sep after A_I1 A_O1 S1 when randomized constraints follow", "TranscriptProtocol::r1cs_2phase_domain_sep"),
    L("r1cs.V", "merlin-label", ["V"], "append_point: each high-level commitment, in commit order", "r1cs::Prover::commit / Verifier::commit"),
    L("r1cs.m", "merlin-label", ["m"], "append_u64: number of commitments, before the first-phase points", "r1cs::Verifier::verify"),
    L("r1cs.phase1", "merlin-label", ["A_I1", "A_O1", "S1"], "first-phase vector commitments (identity rejected: validate_and_append_point)", "r1cs::Verifier::verify"),
    L("r1cs.phase2", "merlin-label", ["A_I2", "A_O2", "S2"], "second-phase vector commitments (append_point: the identity is allowed -- one-phase proofs carry it)", "r1cs::Verifier::verify"),
    L("r1cs.yz", "merlin-label", ["y", "z"], "challenge_scalar y then z", "r1cs::Verifier::verify"),
    L("r1cs.T", "merlin-label", ["T_1", "T_3", "T_4", "T_5", "T_6"], "polynomial commitments (identity rejected); there is no T_2", "r1cs::Verifier::verify"),
    L("r1cs.ux", "merlin-label", ["u", "x"], "challenge_scalar u then x ('u' is reused by the inner-product rounds)", "r1cs::Verifier::verify"),
    L("r1cs.tx", "merlin-label", ["t_x", "t_x_blinding", "e_blinding"], "append_scalar of the three proof scalars", "r1cs::Verifier::verify"),
    L("r1cs.w", "merlin-label", ["w"], "challenge_scalar w: folds <l,r> into the inner-product statement", "r1cs::Verifier::verify"),
    L("r1cs.v_blinding", "merlin-label", ["v_blinding"], "prover only: rekey_with_witness_bytes of each commitment's blinding into the TranscriptRng", "r1cs::Prover::prove"),
    L("ipp.domsep", "domain-separator", ["ipp v1"], "dom-sep of the inner-product argument", "TranscriptProtocol::innerproduct_domain_sep"),
    L("ipp.n", "merlin-label", ["n"], "append_u64: padded vector length, after the ipp dom-sep (also: key count of Musig.aggregated-key)", "TranscriptProtocol::innerproduct_domain_sep"),
    L("ipp.LR", "merlin-label", ["L", "R"], "per round: L_j then R_j (identity rejected), then challenge 'u' (also: children of a Merkle node, and R of the signature)", "InnerProductProof::verification_scalars"),
    # ---- randomized-constraint challenges of the cloak gadget (spacesuit) ---------------------------------------------------
    L("cloak.shuffle", "merlin-label", ["shuffle challenge"], "challenge of scalar_shuffle", "spacesuit::shuffle::scalar_shuffle"),
    L("cloak.kvalue", "merlin-label", ["k-value shuffle challenge"], "challenge of value_shuffle", "spacesuit::shuffle::value_shuffle"),
    L("cloak.mix", "merlin-label", ["mix challenge"], "challenge of the 2-in/2-out mix gadget", "spacesuit::mix"),
    L("zkvm.r1cs", "transcript-label", ["ZkVM.r1cs"], "Transcript::new label of the transaction's constraint system", "zkvm::vm::VM::run / Verifier::verify_tx"),
    # ---- generators ---------------------------------------------------------------------------------------------------
    L("gens.chain", "generator-label", ["GeneratorsChain"], "SHAKE256 input prefix, followed by 'G' or 'H' and LE32(party); 64 bytes per point through from_uniform_bytes",
      "bulletproofs::generators::GeneratorsChain::new"),
    # ---- transaction format (zkvm) ----------------------------------------------------------------------------------------
    L("tx.contractid", "transcript-label", ["ZkVM.contractid"], "Transcript::new label of a contract id", "zkvm::contract::Contract::id"),
    L("tx.contractid.labels", "merlin-label", ["contract", "id"], "message 'contract' (serialized contract), challenge_bytes 'id' (32)", "zkvm::contract::Contract::id"),
    L("tx.ratchet", "transcript-label", ["ZkVM.ratchet-anchor"], "Transcript::new label of Anchor::ratchet", "zkvm::contract::Anchor::ratchet"),
    L("tx.ratchet.labels", "merlin-label", ["old", "new"], "message 'old' (32), challenge_bytes 'new' (32)", "zkvm::contract::Anchor::ratchet"),
    L("tx.txid", "transcript-label", ["ZkVM.txid"], "label of the Merkle hasher over the transaction log", "zkvm::tx::TxID::from_log / zkvm::merkle::MerkleTree"),
    L("tx.header.labels", "merlin-label", ["tx.version", "tx.mintime", "tx.maxtime"], "append_u64 fields of the Header log entry", "zkvm::tx::TxEntry::Header (MerkleItem::commit)"),
    L("tx.entry.labels", "merlin-label", ["input", "output"], "message label of an Input / Output log entry (32-byte contract id)", "zkvm::tx::TxEntry (MerkleItem::commit)"),
    L("tx.merkle.labels", "merlin-label", ["merkle.leaf", "merkle.node", "merkle.empty"], "challenge_bytes labels of leaf / inner node / empty tree", "zkvm::merkle::MerkleTree"),
    L("tx.musig", "transcript-label", ["Musig.aggregated-key"], "Transcript::new label of the multikey aggregation", "musig::Multikey::new"),
    L("tx.musig.labels", "merlin-label", ["X", "i", "a_i"], "append_point 'X' per key (after 'n'); per key a clone with append_u64 'i', challenge_scalar 'a_i' (also: 'X' = aggregated key of the signature)", "musig::Multikey::new"),
    L("tx.signtx", "transcript-label", ["ZkVM.signtx"], "Transcript::new label of the transaction signature", "zkvm::vm::VM::run (signtx) / Signature::verify"),
    L("tx.signtx.labels", "merlin-label", ["txid", "c"], "message 'txid' (32) before the signature's own transcript; challenge_scalar 'c'", "zkvm::vm / musig::Signature::verify"),
    L("tx.schnorr.domsep", "domain-separator", ["schnorr-signature v1"], "dom-sep of the Schnorr signature, then 'X', 'R', challenge 'c'", "musig::transcript::TranscriptProtocol::schnorr_sig_domain_sep"),
]

# literals of the strict files that are NOT restatements of upstream (each with the reason an integrator can skip it)
OWN = {
    "unused": "placeholder label of a Transcript member that is overwritten before use (r1cs_prover.hpp)",
    "q_blinding": "this repository's deterministic derivation of the quantity blinding from a seed (upstream draws it from an RNG)",
    "f_blinding": "the same for the flavor blinding",
    "blinding": "the same for described systems",
    "tx": "seed derivation of the synthetic transaction generator (oracle/cloak.c)",
    "flavor": "the same", "amount": "the same", "anchor": "the same (oracle/zkvm_tx.c builder)", "key": "the same", "recipient": "the same", "nonce": "the same",
    "": "empty label slot 0 of the hash-plan label table (zkvm_tx.hpp)",
    "zkvm_amd.gadget": "transcript label of this repository's own test statements (range / shuffle described as data); no upstream counterpart",
}

# non-literal recollections: found by a regular expression on each side; `what` is the claim an integrator checks
STRUCTURE = [
    {"id": "proof.version1", "kind": "wire-format",
     "what": "R1CSProof bytes, two-phase: version byte 0x01 | A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 (11 x 32) | t_x t_x_blinding e_blinding (3 x 32) | L_0 R_0 .. L_{k-1} R_{k-1} (2k x 32) | a b (2 x 32); total 1 + 32 (16 + 2k)",
     "upstream": "bulletproofs::r1cs::R1CSProof::to_bytes / from_bytes",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"len < 1 \+ 32 \* 16 \|\| proof\[0\] != 1"), ("zkvm_amd/csrc/prep_kernels.hpp", r"version byte and length must agree"),
                 ("zkvm_amd/csrc/transcript_tape.hpp", r'append_data\("t_x", TAPE_SRC_PROOF, 32 \* 11'), ("zkvm_amd/csrc/prover_dev.hpp", r"proof_bytes\[0\] = 1;")],
     "oracle": [("oracle/r1cs.c", r"proof_len < 1 \+ 32 \* 16 \|\| proof\[0\] != 1"), ("oracle/r1cs.c", r"const uint8_t \*scb = pt \+ 32 \* 11;")]},
    {"id": "proof.version0", "kind": "wire-format",
     "what": "R1CSProof bytes, one-phase: version byte 0x00 and A_I2 A_O2 S2 left out (13 + 2k elements); read as the two-phase form with the identity in their place; a version byte that disagrees with the length is malformed",
     "upstream": "bulletproofs::r1cs::R1CSProof::from_bytes (missing_phase2_commitments)",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"len >= 1 \+ 32 \* 13 && proof\[0\] == 0"), ("zkvm_amd/csrc/zkgpu.hip", r"inline bool proof_len_fits")],
     "oracle": [("oracle/r1cs.c", r"proof_len >= 1 \+ 32 \* 13 && proof\[0\] == 0")]},
    {"id": "proof.identity_rules", "kind": "behaviour",
     "what": "the identity encoding (32 zero bytes) is rejected for A_I1 A_O1 S1, T_1 T_3 T_4 T_5 T_6 and every L_j R_j (validate_and_append_point), allowed for A_I2 A_O2 S2 (append_point)",
     "upstream": "bulletproofs::transcript::TranscriptProtocol::validate_and_append_point, r1cs::Verifier::verify",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"if \(is_identity\(pt\) \|\| is_identity\(pt \+ 32\)"), ("zkvm_amd/csrc/r1cs_verifier.hpp", r"for \(int i = 6; i < 11; \+\+i\) if \(is_identity"),
                 ("zkvm_amd/csrc/prep_kernels.hpp", r"well-formedness: no identity among the proof points")],
     "oracle": [("oracle/r1cs.c", r"#define VALIDATE\(ptr\)")]},
    {"id": "msm.layout", "kind": "verification-equation",
     "what": "one multiscalar multiplication == identity over [A_I1 A_O1 S1 A_I2 A_O2 S2 | V_j | T_1 T_3 T_4 T_5 T_6 | B B_blinding | G_i | H_i | L_j R_j] with scalars [x x^2 x^3 u x u x^2 u x^3 | r x^2 wV_j | r x, r x^3 .. r x^6 | w(t_x - a b) + r(x^2 (wc + delta) - t_x), -e_blinding - r t_x_blinding | g_i | h_i | u_j^2 u_j^-2] (r = the verifier's random weight)",
     "upstream": "bulletproofs::r1cs::Verifier::verify (mega_check)",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"struct VerifierMsm")],
     "oracle": [("oracle/r1cs.c", r"IPA verification scalars")]},
    {"id": "pedersen.B_blinding", "kind": "generator",
     "what": "B = ristretto255 basepoint; B_blinding = from_uniform_bytes(SHA3-512(compress(B)))  [this one IS pinned: dalek's published constant 8c9240b4...48871134, tests/test_oracle_group.py]",
     "upstream": "bulletproofs::PedersenGens::default",
     "product": [("zkvm_amd/csrc/zkgpu.hip", r"Sponge sp = sha3_512_sponge\(\);")],
     "oracle": [("oracle/merlin.c", r"^void pedersen_gens\(")]},
    {"id": "gens.chain.layout", "kind": "generator",
     "what": "G_i / H_i of party j = from_uniform_bytes of the i-th 64-byte block of SHAKE256('GeneratorsChain' || 'G' or 'H' || LE32(j)); BulletproofGens::new(capacity, 1): party 0 only",
     "upstream": "bulletproofs::generators::{GeneratorsChain, BulletproofGens::new}",
     "product": [("zkvm_amd/csrc/zkgpu.hip", r"\(uint8_t\)\(side \? 'H' : 'G'\)")],
     "oracle": [("oracle/merlin.c", r"uint8_t label\[5\] = \{\(uint8_t\)which")]},
    {"id": "challenge.wide", "kind": "behaviour",
     "what": "challenge_scalar = Scalar::from_bytes_mod_order_wide of 64 challenge bytes; append_u64 = 8 bytes little-endian; append_point / append_scalar = the 32-byte encoding",
     "upstream": "bulletproofs::transcript::TranscriptProtocol",
     "product": [("zkvm_amd/csrc/merlin.hpp", r"Scalar challenge_scalar\(const char\* label\)")],
     "oracle": [("oracle/merlin.c", r"void merlin_challenge_scalar")]},
    {"id": "cloak.var_order", "kind": "gadget-layout",
     "what": "cloak(in, out): commitments (q, f) per value, inputs first (committed variable 2i = quantity, 2i+1 = flavor); gadget order k_mix(in) -> k_mix(out) -> value_shuffle(in, merge_in) -> padded_shuffle(merge_out, split_in) -> value_shuffle(split_out, out) -> range_proof(out.q, 64 bits); k_mix allocates grouped (k), mid (k-2), merged (k) in that order; a 2-in/2-out cloak has 150 multipliers (padded 256), 8 commitments",
     "upstream": "spacesuit::cloak::cloak, spacesuit::{mix::k_mix, shuffle, range_proof}",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"void gadget\(CS& cs, const std::vector<Value>& in"), ("zkvm_amd/csrc/r1cs_verifier.hpp", r"void k_mix\(CS& cs")],
     "oracle": [("oracle/cloak.c", r"int rc = k_mix\(cs, in, n_in, merge_in, merge_out\);"), ("oracle/cloak.c", r"^static int k_mix\(")]},
    {"id": "cloak.mix_equation", "kind": "gadget-layout",
     "what": "mix(A, B, C, D) with challenge w: one multiplier (A.q - C.q + w(A.f - C.f) + w^2(B.q - D.q) + w^3(B.f - D.f)) * (C.q + w^4(A.f - B.f) + w^2(D.q - A.q - B.q) + w^3(D.f - A.f)) = 0",
     "upstream": "spacesuit::mix::mix",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"const S w2 = w \* w, w3 = w2 \* w, w4 = w3 \* w")],
     "oracle": [("oracle/cloak.c", r"mix challenge")]},
    {"id": "cloak.range_bits", "kind": "gadget-layout",
     "what": "range proof of every OUTPUT quantity over 64 bits: per bit a multiplier (a, b, o) with o = 0, a + b - 1 = 0, and q - sum b_i 2^i = 0",
     "upstream": "spacesuit::range_proof::range_proof",
     "product": [("zkvm_amd/csrc/r1cs_verifier.hpp", r"range_proof\(cs, o\.q, 64\)")],
     "oracle": [("oracle/cloak.c", r"range_proof\(cs, out\[i\]\.q, .*64\);")]},
    {"id": "tx.layout", "kind": "wire-format",
     "what": "Tx := version:u64 | mintime_ms:u64 | maxtime_ms:u64 | n:u32 program[n] | R:32 s:32 | n:u32 R1CSProof[n]   (all little-endian; DESIGN.md sec 4.5)",
     "upstream": "zkvm::tx::Tx::{encode, decode}",
     "product": [("zkvm_amd/csrc/zkvm_tx.hpp", r"if \(len < 24 \+ 4\)")],
     "oracle": [("oracle/zkvm_tx.c", r"out->version = le64\(tx\); out->mintime = le64\(tx \+ 8\)")]},
    {"id": "tx.contract", "kind": "wire-format",
     "what": "Contract := anchor:32 | predicate:32 | k:u32 | item*;  item := 0x00 n:u32 bytes[n] (string) | 0x02 qty:32 flavor:32 (value)",
     "upstream": "zkvm::contract::{Contract, PortableItem}::encode",
     "product": [("zkvm_amd/csrc/zkvm_tx.hpp", r"malformed contract")],
     "oracle": [("oracle/zkvm_tx.c", r"Contract := anchor:32 \| predicate:32")]},
    {"id": "tx.opcodes", "kind": "opcode",
     "what": "0x00 push:n:x  0x02 drop  0x03 dup:k  0x04 roll:k  0x06 var  0x18 cloak:m:n  0x1b input  0x1c output:k  0x20 signtx  (immediates u32 LE); any other opcode: outside the subset (status 2), never 'invalid'",
     "upstream": "zkvm::ops::Opcode",
     "product": [("zkvm_amd/csrc/zkvm_tx.hpp", r"case 0x18:"), ("zkvm_amd/csrc/zkvm_tx.hpp", r"case 0x1b:"), ("zkvm_amd/csrc/zkvm_tx.hpp", r"case 0x1c:"), ("zkvm_amd/csrc/zkvm_tx.hpp", r"case 0x20:")],
     "oracle": [("oracle/zkvm_tx.c", r"case 0x18:"), ("oracle/zkvm_tx.c", r"case 0x1b:"), ("oracle/zkvm_tx.c", r"case 0x1c:"), ("oracle/zkvm_tx.c", r"case 0x20:")]},
    {"id": "tx.txid.tree", "kind": "behaviour",
     "what": "transaction ID = Merkle root (RFC 6962 split: largest power of two below the count) over [header, inputs and outputs in program order]; leaf = Merlin('ZkVM.txid') + entry fields + challenge 'merkle.leaf'; node = 'L' 'R' + 'merkle.node'; empty = 'merkle.empty'",
     "upstream": "zkvm::merkle::MerkleTree::root, zkvm::tx::TxID::from_log",
     "product": [("zkvm_amd/csrc/zkvm_tx.hpp", r'"merkle\.node"')],
     "oracle": [("oracle/zkvm_tx.c", r"merkle\.node")]},
    {"id": "tx.signature.equation", "kind": "verification-equation",
     "what": "X = sum a_i X_i with a_i from 'Musig.aggregated-key' (n, X_1..X_n, then per key i: 'i', challenge 'a_i'); c from 'ZkVM.signtx' (txid, dom-sep 'schnorr-signature v1', X, R); accept iff s B = R + c X",
     "upstream": "musig::{Multikey::new, Signature::verify}, zkvm::vm signtx",
     "product": [("zkvm_amd/csrc/zkvm_tx.hpp", r"inline void tx_finish_signature\(")],
     "oracle": [("oracle/zkvm_tx.c", r"schnorr-signature v1")]},
]

# the label sequence of r1cs::Verifier::verify for a TWO-phase proof with m commitments and k rounds, as (op, label):
# extracted from three places and compared with this list by the test
R1CS_SEQUENCE = ["dom-sep", "V", "m", "A_I1", "A_O1", "S1", "dom-sep", "A_I2", "A_O2", "S2", "y", "z", "T_1", "T_3", "T_4", "T_5", "T_6",
                 "u", "x", "t_x", "t_x_blinding", "e_blinding", "w", "dom-sep", "n", "L", "R", "u"]


def strip_comments(text: str, path: str) -> str:
    """comments -> spaces, newlines kept (line numbers stay true)"""
    keep_nl = lambda m: re.sub(r"[^\n]", " ", m.group(0))                  # noqa: E731
    if path.endswith(".py"):
        return re.sub(r"#[^\n]*", keep_nl, text)
    text = re.sub(r"/\*.*?\*/", keep_nl, text, flags=re.S)
    return re.sub(r"//[^\n]*", keep_nl, text)


def literals(path: str):
    """-> [(line number, literal)] of the short string literals in the code of a file"""
    text = strip_comments(open(os.path.join(ROOT, path), errors="replace").read(), path)
    out = []
    for no, line in enumerate(text.split("\n"), 1):
        if NOT_CODE.search(line):
            continue
        for m in re.finditer(r'"((?:[^"\\\n]|\\.){0,40})"', line):
            out.append((no, m.group(1)))
    return out


def find_literal(files, value):
    where = []
    for f in files:
        for no, lit in literals(f):
            if lit == value:
                where.append("%s:%d" % (f, no))
    return where


def find_regex(spec):
    out = []
    for f, rx in spec:
        text = open(os.path.join(ROOT, f), errors="replace").read().split("\n")
        hits = [i + 1 for i, line in enumerate(text) if re.search(rx, line)]
        out.append({"file": f, "regex": rx, "lines": hits[:6]})
    return out


def build():
    labels = []
    for e in CATALOGUE:
        item = dict(e)
        item["product"] = {v: find_literal(PRODUCT_STRICT + PRODUCT_EXTRA, v) for v in e["values"]}
        item["oracle"] = {v: find_literal(ORACLE_STRICT + ORACLE_EXTRA, v) for v in e["values"]}
        labels.append(item)
    structure = []
    for s in STRUCTURE:
        item = {k: v for k, v in s.items() if k not in ("product", "oracle")}
        item["product"] = find_regex(s["product"])
        item["oracle"] = find_regex(s["oracle"])
        structure.append(item)
    return {
        "_about": "Every constant of this repository that restates upstream from memory and that nothing held here pins. Generated by "
                  "tools/unpinned_manifest.py (catalogue by hand, locations by scanning); checked by tests/test_unpinned_manifest.py. "
                  "An integrator with the real sources diffs THIS file against them.",
        "reference_anchor": "/root/reference/README.md:3-6 (the repository has moved; no source is mounted) -- SURVEY.md sec 0, sec 8(c), sec 8(f-1)",
        "strict_files": {"product": PRODUCT_STRICT, "oracle": ORACLE_STRICT},
        "labels": labels,
        "own_literals": OWN,
        "structure": structure,
        "r1cs_verifier_label_sequence": R1CS_SEQUENCE,
    }


MD = os.path.join(ROOT, "oracle", "UNPINNED.md")


def markdown(m) -> str:
    """the same manifest for people: oracle/UNPINNED.md"""
    def locs(d):
        seen = []
        for v in d.values():
            for x in v:
                if x not in seen:
                    seen.append(x)
        return ", ".join("`%s`" % x.replace("zkvm_amd/csrc/", "").replace("oracle/", "") for x in seen[:8]) + (" …" if len(seen) > 8 else "")
    out = ["# UNPINNED — what this repository restates from memory, and where",
           "",
           "Generated by `tools/unpinned_manifest.py` from its catalogue and a scan of the sources; the machine-readable twin is",
           "`tests/golden/unpinned_manifest.json`, kept true by `tests/test_unpinned_manifest.py`. **Parity unpinned**: `/root/reference`",
           "holds no source and no vector (`README.md:3-6`), so every row below is a RECOLLECTION of upstream (slingshot `zkvm`,",
           "`spacesuit`, dalek `bulletproofs` with `yoloproofs`, `merlin`, `musig`) that four implementations here agree on byte for byte",
           "(oracle prover, oracle verifier, product host verifier, product device path) and that nothing here can hold against",
           "upstream. Whoever has the real sources diffs these rows against them; vectors go to `tests/golden/upstream/` (schema in its",
           "README) and are then checked by `tests/test_upstream_vectors.py` on CPU (oracle) and under `-m gpu` (HIP path).",
           "",
           "Product paths are under `zkvm_amd/csrc/`, oracle paths under `oracle/`.",
           "",
           "## Labels, domain separators, protocol constants",
           "",
           "| id | value(s) | role | upstream item (recollected) | pinned by | product | oracle |",
           "|---|---|---|---|---|---|---|"]
    for e in m["labels"]:
        out.append("| %s | %s | %s | `%s` | %s | %s | %s |" % (e["id"], " ".join("`%s`" % v for v in e["values"]), e["role"], e["upstream"], e["pinned_by"],
                                                            locs(e["product"]), locs(e["oracle"])))
    out += ["", "Label order of `r1cs::Verifier::verify` (two-phase proof; `V` once per commitment, `L R u` once per round):",
            "`" + " → ".join(m["r1cs_verifier_label_sequence"]) + "`", "",
            "## Wire formats, layouts, equations, behaviour", "",
            "| id | kind | the claim to check | upstream item (recollected) | product | oracle |", "|---|---|---|---|---|---|"]
    for s_ in m["structure"]:
        def where(side):
            return ", ".join("`%s:%s`" % (a["file"].replace("zkvm_amd/csrc/", "").replace("oracle/", ""), "/".join(str(x) for x in a["lines"][:3])) for a in s_[side])
        out.append("| %s | %s | %s | `%s` | %s | %s |" % (s_["id"], s_["kind"], s_["what"].replace("|", "\\|"), s_["upstream"], where("product"), where("oracle")))
    out += ["", "## Literals in the protocol sources that are this repository's own (nothing to diff)", ""]
    for k, v in m["own_literals"].items():
        out.append("- `%s` — %s" % (k, v))
    return "\n".join(out) + "\n"


def main():
    m = build()
    doc = json.dumps(m, indent=1, sort_keys=False) + "\n"
    md = markdown(m)
    if "--check" in sys.argv:
        have = open(OUT).read() if os.path.exists(OUT) else ""
        have_md = open(MD).read() if os.path.exists(MD) else ""
        if have != doc or have_md != md:
            print("tests/golden/unpinned_manifest.json / oracle/UNPINNED.md are stale: run python tools/unpinned_manifest.py", file=sys.stderr)
            sys.exit(1)
        return
    with open(OUT, "w") as f:
        f.write(doc)
    with open(MD, "w") as f:
        f.write(md)
    print(OUT)
    print(MD)


if __name__ == "__main__":
    main()
