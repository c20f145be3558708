"""tests/golden/unpinned_manifest.json lists every constant that restates upstream from memory, with file:line in the
product AND in the oracle.  These tests keep it honest: it is what scanning the sources gives; no
label literal of a protocol source file on either side is missing from it; every entry exists on both sides; and the label
ORDER of r1cs::Verifier::verify is one and the same in the manifest, the product's host verifier, the product's device
tape and the oracle."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import unpinned_manifest as um  # noqa: E402


def _manifest():
    return json.load(open(um.OUT))


def test_committed_manifest_is_what_the_sources_give():
    m = um.build()
    assert json.dumps(m, indent=1) + "\n" == open(um.OUT).read(), "run: python tools/unpinned_manifest.py"
    assert um.markdown(m) == open(um.MD).read(), "oracle/UNPINNED.md is stale: run python tools/unpinned_manifest.py"


def test_every_label_literal_of_the_protocol_sources_is_classified():
    m = _manifest()
    known = {v for e in m["labels"] for v in e["values"]} | set(m["own_literals"])
    for side in ("product", "oracle"):
        for f in m["strict_files"][side]:
            for no, lit in um.literals(f):
                assert lit in known, "%s:%d: the literal %r is neither in the unpinned manifest nor in its list of the repository's own literals" % (f, no, lit)


def test_every_entry_is_found_on_both_sides_with_its_lines():
    m = _manifest()
    assert len(m["labels"]) >= 35 and len(m["structure"]) >= 15
    for e in m["labels"]:
        assert e["upstream"] and e["role"] and e["pinned_by"]
        for v in e["values"]:
            for side in ("product", "oracle"):
                where = e[side][v]
                assert where, (e["id"], v, "not found in the " + side)
                for loc in where:
                    f, no = loc.rsplit(":", 1)
                    line = open(os.path.join(ROOT, f), errors="replace").read().split("\n")[int(no) - 1]
                    assert '"%s"' % v in line, (loc, v)
    for s in m["structure"]:
        assert s["what"] and s["upstream"]
        for side in ("product", "oracle"):
            assert s[side], (s["id"], side)
            for a in s[side]:
                assert a["lines"], (s["id"], side, a["file"], a["regex"], "anchor not found")


def _sequence(path, start, stop, vocabulary):
    """label literals in order of appearance between two anchor lines, consecutive repeats folded"""
    text = um.strip_comments(open(os.path.join(ROOT, path), errors="replace").read(), path).split("\n")
    lo = next(i for i, l in enumerate(text) if re.search(start, l))
    hi = next(i for i, l in enumerate(text) if i > lo and re.search(stop, l))
    seq = []
    for line in text[lo:hi]:
        found = [(m.start(), m.group(1)) for m in re.finditer(r'"((?:[^"\\\n]|\\.){0,40})"', line) if m.group(1) in vocabulary]
        for _, tok in sorted(found):
            if not seq or seq[-1] != tok:
                seq.append(tok)
    return seq


def test_r1cs_verifier_label_order_is_the_same_in_manifest_product_tape_and_oracle():
    want = _manifest()["r1cs_verifier_label_sequence"]
    vocab = set(want)
    host = _sequence("zkvm_amd/csrc/r1cs_verifier.hpp", r"explicit R1csVerifier\(const char\* label\)", r"ch\[j\] = tr_\.challenge_scalar", vocab)
    host.append("u")                                       # (the stop line itself: the round's challenge)
    assert host == want, host
    tape = _sequence("zkvm_amd/csrc/transcript_tape.hpp", r'rec\.append_data\("V", TAPE_SRC_COMMITMENTS', r"rec\.challenge\(\"u\", ch_fixed", vocab)
    tape.append("u")
    assert tape == want[1:], tape                          # (the tape starts after Transcript::new + "r1cs v1": the initial state is computed on the host)
    head = _sequence("oracle/r1cs.c", r"^static r1cs_cs \*cs_new", r"^r1cs_var r1cs_verifier_commit", vocab)
    # oracle/r1cs.c holds the prover's "m" line first: take the verifier's (the second)
    text = open(os.path.join(ROOT, "oracle/r1cs.c")).read()
    assert text.count('merlin_append_u64(tr, "m", cs->m);') == 2
    second = text.index('merlin_append_u64(tr, "m", cs->m);', text.index('merlin_append_u64(tr, "m", cs->m);') + 1)
    tail = text[second:text.index("sc one, zero, allinv;", second)]
    seq = []
    for m in re.finditer(r'"((?:[^"\\\n]|\\.){0,40})"|create_randomized_constraints\(cs\)', tail):
        tok = "dom-sep" if m.group(0).startswith("create") else m.group(1)
        if tok in vocab and (not seq or seq[-1] != tok):
            seq.append(tok)
    assert head == ["dom-sep", "V"], head
    assert head + seq == want, head + seq
