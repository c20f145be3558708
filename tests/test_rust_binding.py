"""The Rust side of the boundary, checked mechanically (north_star: "host code stays Rust and calls
hand-written HIP kernels through a thin C-ABI FFI layer").  There is no rustc / cargo in this image, so
the compile check a binding can have here is this one: both sides parsed (tests/abi_util.py) into one canonical form and
compared -- names, arity, integer width and signedness, pointer depth, constness of every pointer level, struct fields,
constants -- plus the library's dynamic symbol table, plus every call the safe wrapper makes into the raw crate."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from abi_util import parse_c_header, parse_rust_sys, split_top  # noqa: E402

HEADER = os.path.join(ROOT, "include", "zkgpu.h")
SYS = os.path.join(ROOT, "rust", "zkgpu-sys", "src", "lib.rs")
SAFE = os.path.join(ROOT, "rust", "zkgpu", "src", "lib.rs")


@pytest.fixture(scope="module")
def both():
    c_funcs, c_structs = parse_c_header(open(HEADER).read())
    r_funcs, r_structs, r_opaque = parse_rust_sys(open(SYS).read())
    return c_funcs, c_structs, r_funcs, r_structs, r_opaque


def test_every_export_of_the_header_is_declared_in_rust_with_the_same_machine_types(both):
    c_funcs, _, r_funcs, _, _ = both
    assert 80 <= len(c_funcs) <= 90          # (the 25 hooks of zkgpu_hooks.h are neither exported nor bound)
    assert sorted(c_funcs) == sorted(r_funcs)                       # no export missing, none invented
    for name, (c_ret, c_args) in c_funcs.items():
        r_ret, r_args = r_funcs[name]
        assert c_ret == r_ret, (name, "return", c_ret, r_ret)
        assert len(c_args) == len(r_args), (name, "arity", len(c_args), len(r_args))
        for i, ((cn, ct), (rn, rt)) in enumerate(zip(c_args, r_args)):
            assert ct == rt, (name, i, cn, ct, rt)


def test_the_parsers_see_a_deliberate_mismatch():
    """the comparison above is only worth something if a wrong declaration fails it"""
    c_funcs, _ = parse_c_header("int zkgpu_x(zkgpu_ctx *c, const uint8_t *p, size_t n, uint64_t *out, const void *const *q, uint8_t id[128]);")
    good = 'extern "C" {\n    pub fn zkgpu_x(c: *mut zkgpu_ctx, p: *const u8, n: usize, out: *mut u64, q: *const *const c_void, id: *mut u8) -> c_int;\n}'
    assert parse_rust_sys(good)[0] == c_funcs
    for wrong in (good.replace("n: usize", "n: u32"), good.replace("p: *const u8", "p: *mut u8"), good.replace("out: *mut u64", "out: u64"),
                  good.replace("*const *const c_void", "*const *mut c_void"), good.replace("-> c_int", "-> u64"),
                  good.replace("id: *mut u8", "id: *mut u8, extra: c_int")):
        assert parse_rust_sys(wrong)[0] != c_funcs


def test_struct_layouts_opaque_handles_and_constants(both):
    c_funcs, c_structs, _, r_structs, r_opaque = both
    assert list(c_structs) == ["zkgpu_r1cs_desc"]
    assert [(n, t) for n, t in c_structs["zkgpu_r1cs_desc"]] == [(n, t) for n, t in r_structs["zkgpu_r1cs_desc"]]      # same fields, same order
    handles = {t[-1] for ret, args in c_funcs.values() for t in [ret] + [a[1] for a in args] if t[-1].startswith("zkgpu_")} - set(c_structs)
    assert sorted(handles) == sorted(r_opaque)
    header, rust = open(HEADER).read(), open(SYS).read()
    consts = dict(re.findall(r"#define\s+(ZKGPU_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", header))
    consts.pop("ZKGPU_H", None)
    assert len(consts) >= 9
    for name, val in consts.items():
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % name, rust)
        assert m and m.group(1) == val, name


def test_the_library_exports_exactly_what_both_sides_declare(both):
    from zkvm_amd import build
    build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", build.OUT], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("zkgpu_"))
    assert exported == sorted(both[0])


def test_committed_declarations_are_what_the_generator_gives_for_the_current_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "rust/zkgpu-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py"


def _calls(src):
    """(function, number of arguments) of every `sys::zkgpu_*(...)` call in the safe wrapper"""
    src = re.sub(r"//[^\n]*", " ", src)
    out = []
    for m in re.finditer(r"\bsys::(zkgpu_\w+)\s*\(", src):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        inner = src[m.end(): i - 1]
        out.append((m.group(1), len([a for a in split_top(inner) if a.strip()])))
    return out


def test_safe_wrapper_calls_exist_with_the_right_number_of_arguments(both):
    """rust/zkgpu (GpuVerifier::{new, verify_txs, verify_block, submit, wait}, Comm, Prover): every call into the raw crate
    names a declared function and passes as many arguments as it takes; the entry points INTEGRATION.md sec 3 describes exist."""
    r_funcs = both[2]
    src = open(SAFE).read()
    calls = _calls(src)
    assert len(calls) >= 25
    for name, n in calls:
        assert name in r_funcs, name
        assert n == len(r_funcs[name][1]), (name, n, len(r_funcs[name][1]))
    for needed in ("pub struct GpuVerifier", "pub fn new(", "pub fn verify_txs(", "pub fn verify_block(", "pub fn submit(", "pub fn wait(",
                   "impl Drop for GpuVerifier", "pub enum Error"):
        assert needed in src, needed
    used = {n for n, _ in calls}
    for core in ("zkgpu_init", "zkgpu_destroy", "zkgpu_verifier_create", "zkgpu_verifier_destroy", "zkgpu_verifier_verify",
                 "zkgpu_tx_verify_batch", "zkgpu_verifier_set_tx_format", "zkgpu_verifier_submit", "zkgpu_verifier_wait",
                 "zkgpu_pointset_build_tables", "zkgpu_comm_create", "zkgpu_verifier_verify_sharded"):
        assert core in used, core
    # braces and parentheses balance (the cheapest syntax check there is)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', re.sub(r"//[^\n]*", "", src))
    code = re.sub(r"'(?:[^'\\]|\\.)'", "' '", code)
    for a, b in ("{}", "()", "[]"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))
