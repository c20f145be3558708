#!/usr/bin/env python3
"""Writes rust/zkgpu-sys/src/lib.rs from include/zkgpu.h -- what `bindgen` would do; neither bindgen nor cargo exists in this
image, so the declarations are produced with the small C parser the ABI test uses (tests/abi_util.py) and
committed as source.  tests/test_rust_binding.py parses BOTH files again, independently of this script, and fails on any
name, arity, integer-width, pointer-depth or constness mismatch: run this after every change of the header.

    python tools/gen_rust_sys.py            # rewrite the file
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file is not what the header gives
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from abi_util import parse_c_header  # noqa: E402

RUST = {"i32": "c_int", "u32": "u32", "u64": "u64", "i64": "c_longlong", "usize": "usize", "u8": "u8", "char": "c_char",
        "void": "c_void", "f64": "f64", "u16": "u16"}
KEYWORDS = {"type", "in", "ref", "box", "move", "mod", "fn", "loop", "match", "self", "dyn", "final", "override", "priv", "where"}


def rust_ty(t):
    out = ""
    for tok in t[:-1]:
        out += "*const " if tok == "ptr_const" else "*mut "
    return out + RUST.get(t[-1], t[-1])


def main():
    header = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    funcs, structs = parse_c_header(header)
    order = [m.group(1) for m in re.finditer(r"\b(zkgpu_\w+)\s*\(", re.sub(r"/\*.*?\*/", " ", header, flags=re.S))]
    seen, names = set(), []
    for n in order:
        if n in funcs and n not in seen:
            seen.add(n)
            names.append(n)
    consts = re.findall(r"#define\s+(ZKGPU_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", header)
    opaque = sorted({t[-1] for ret, args in funcs.values() for t in [ret] + [a[1] for a in args]
                     if t[-1].startswith("zkgpu_") and t[-1] not in structs})
    o = []
    o.append("//! Raw FFI declarations of `libzkgpu.so` (include/zkgpu.h, ABI version 3): the MI355X back end of the ZkVM /")
    o.append("//! Bulletproofs-R1CS verification path.  One declaration per exported function, in the header's order; the")
    o.append("//! header's comments are the documentation (conventions, ownership, fail-closed rules) and are not repeated here.")
    o.append("//!")
    o.append("//! Produced from the header by `tools/gen_rust_sys.py` and committed as source; `tests/test_rust_binding.py`")
    o.append("//! re-parses both files and fails on any mismatch of name, arity, integer width, pointer depth or constness.")
    o.append("//! NOT COMPILED in the build container (no rustc / cargo there): a maintainer runs `cargo check` first.")
    o.append("#![allow(non_camel_case_types, non_snake_case)]")
    o.append("")
    o.append("use std::os::raw::{c_char, c_int, c_longlong, c_void};")
    o.append("")
    for name, val in consts:
        if name == "ZKGPU_H":
            continue
        ty = "usize" if name.endswith("_BYTES") else "c_int"
        o.append("pub const %s: %s = %s;" % (name, ty, val))
    o.append("")
    for s in opaque:
        o.append("#[repr(C)]")
        o.append("pub struct %s {" % s)
        o.append("    _private: [u8; 0],")
        o.append("}")
    o.append("/// `typedef zkgpu_cloak_plan zkgpu_r1cs_plan;` -- one type under two names")
    o.append("pub type zkgpu_r1cs_plan = zkgpu_cloak_plan;")
    o.append("")
    for s, fields in structs.items():
        o.append("/// A constraint system handed over as data (see the header for the meaning of every array).")
        o.append("#[repr(C)]")
        o.append("pub struct %s {" % s)
        for n, t in fields:
            o.append("    pub %s: %s," % (n, rust_ty(t)))
        o.append("}")
        o.append("")
    o.append('#[link(name = "zkgpu")]')
    o.append('extern "C" {')
    for n in names:
        ret, args = funcs[n]
        parts = []
        for i, (an, at) in enumerate(args):
            an = an or "arg%d" % i
            if an in KEYWORDS:
                an += "_"
            parts.append("%s: %s" % (an, rust_ty(at)))
        tail = "" if ret == ("void",) else " -> " + rust_ty(ret)
        one = "    pub fn %s(%s)%s;" % (n, ", ".join(parts), tail)
        if len(one) <= 120:
            o.append(one)
        else:
            o.append("    pub fn %s(" % n)
            line = "       "
            for p in parts:
                if len(line) + len(p) + 2 > 118:
                    o.append(line.rstrip())
                    line = "       "
                line += " " + p + ","
            o.append(line.rstrip())
            o.append("    )%s;" % tail)
    o.append("}")
    text = "\n".join(o) + "\n"
    path = os.path.join(ROOT, "rust", "zkgpu-sys", "src", "lib.rs")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text)
    print("%s: %d functions, %d constants, %d opaque types" % (path, len(names), len(consts) - 1, len(opaque)))


if __name__ == "__main__":
    main()
