"""Two small parsers that reduce a declaration to the same canonical form, so that the C header (include/zkgpu.h) and the
Rust `extern "C"` block (rust/zkgpu-sys/src/lib.rs) can be compared mechanically: function name, number of arguments, and
for every argument and the return value its machine type -- integer width and signedness, pointer depth, and the constness of
every pointer level.  There is no Rust toolchain in this image: this comparison is the compile check the
binding can have here.

Canonical type: a tuple of tokens read outermost first, e.g.
    const uint8_t *p     / p: *const u8            -> ("ptr_const", "u8")
    zkgpu_ctx **out      / out: *mut *mut zkgpu_ctx -> ("ptr_mut", "ptr_mut", "zkgpu_ctx")
    const void *const *p / p: *const *const c_void  -> ("ptr_const", "ptr_const", "void")
    uint8_t out[32]      / out: *mut u8             -> ("ptr_mut", "u8")      (an array parameter IS a pointer in C)
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

C_SCALARS = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64",
             "int64_t": "i64", "long long": "i64", "size_t": "usize", "uint8_t": "u8", "char": "char", "void": "void",
             "double": "f64", "float": "f32", "uint16_t": "u16"}
RUST_SCALARS = {"c_int": "i32", "i32": "i32", "u32": "u32", "c_uint": "u32", "u64": "u64", "i64": "i64", "c_longlong": "i64",
                "usize": "usize", "u8": "u8", "c_char": "char", "c_void": "void", "f64": "f64", "f32": "f32", "u16": "u16"}
# typedef zkgpu_cloak_plan zkgpu_r1cs_plan; -- one type under two names in the header
C_ALIASES = {"zkgpu_r1cs_plan": "zkgpu_cloak_plan"}

Signature = Tuple[Tuple[str, ...], List[Tuple[str, Tuple[str, ...]]]]        # (return type, [(argument name, type)])


def strip_c(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    src = re.sub(r'extern\s+"C"\s*\{', " ", src)
    return src


def c_type(decl: str) -> Tuple[str, Tuple[str, ...]]:
    """'const uint8_t *const *name[32]' -> (name, canonical type).  The declarator may have no name (prototypes)."""
    decl = " ".join(decl.replace("*", " * ").split())
    array = False
    m = re.search(r"\[[^\]]*\]\s*$", decl)
    if m:
        array = True
        decl = decl[: m.start()].strip()
    toks = decl.split()
    # base type = everything before the first '*', minus a trailing identifier when there is no '*' (that is the name)
    if "*" in toks:
        star = toks.index("*")
        base, rest = toks[:star], toks[star:]
    else:
        base, rest = toks, []
    name = ""
    if rest:
        if rest[-1] not in ("*", "const"):
            name = rest[-1]
            rest = rest[:-1]
    else:
        known = set(C_SCALARS) | {"const", "struct", "unsigned", "long"}
        if len(base) > 1 and base[-1] not in known and not (base[-2] in ("struct",)):
            name = base[-1]
            base = base[:-1]
    base_const = "const" in base
    base = [t for t in base if t not in ("const", "struct")]
    b = " ".join(base)
    b = C_ALIASES.get(b, b)
    b = C_SCALARS.get(b, b)
    # pointer levels, innermost first: '*' followed by an optional 'const' (constness of THAT pointer)
    levels = []          # constness of what each '*' points to
    pointee_const = base_const
    i = 0
    while i < len(rest):
        assert rest[i] == "*", decl
        levels.append(pointee_const)
        pointee_const = i + 1 < len(rest) and rest[i + 1] == "const"
        i += 2 if pointee_const else 1
    if array:
        levels.append(pointee_const)
    out = tuple("ptr_const" if c else "ptr_mut" for c in reversed(levels)) + (b,)
    return name, out


def parse_c_header(text: str) -> Tuple[Dict[str, Signature], Dict[str, List[Tuple[str, Tuple[str, ...]]]]]:
    """-> ({function: signature}, {struct with a body: [(field, type)]})"""
    src = strip_c(text)
    structs: Dict[str, List[Tuple[str, Tuple[str, ...]]]] = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            if f.strip():
                fields.append(c_type(f.strip()))
        structs[m.group(3)] = fields
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", " ", src)
    funcs: Dict[str, Signature] = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(zkgpu_\w+)\s*\(([^;{}]*)\)\s*;", src, flags=re.S):
        rt = m.group(1).strip()
        ret = ("void",) if rt == "void" else c_type(rt + " _r")[1]
        args_txt = m.group(3).strip()
        args = []
        if args_txt and args_txt != "void":
            for a in args_txt.split(","):
                args.append(c_type(a.strip()))
        funcs[m.group(2)] = (ret, args)
    return funcs, structs


def rust_type(t: str) -> Tuple[str, ...]:
    t = t.strip()
    out = []
    while True:
        m = re.match(r"\*\s*(const|mut)\s+(.*)$", t, flags=re.S)
        if not m:
            break
        out.append("ptr_const" if m.group(1) == "const" else "ptr_mut")
        t = m.group(2).strip()
    t = t.split("::")[-1]
    out.append(RUST_SCALARS.get(t, t))
    return tuple(out)


def split_top(s: str) -> List[str]:
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def parse_rust_sys(text: str) -> Tuple[Dict[str, Signature], Dict[str, List[Tuple[str, Tuple[str, ...]]]], List[str]]:
    """-> ({function: signature}, {#[repr(C)] struct with fields: [(field, type)]}, [opaque struct names])"""
    src = re.sub(r"//[^\n]*", " ", text)
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    funcs: Dict[str, Signature] = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            args = []
            for a in split_top(m.group(2)):
                if not a.strip():
                    continue
                n, ty = a.split(":", 1)
                args.append((n.strip(), rust_type(ty)))
            ret = rust_type(m.group(3)) if m.group(3) else ("void",)
            funcs[m.group(1)] = (ret, args)
    structs: Dict[str, List[Tuple[str, Tuple[str, ...]]]] = {}
    opaque: List[str] = []
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        body = m.group(2)
        fields = []
        for f in split_top(body):
            if ":" not in f:
                continue
            n, ty = f.split(":", 1)
            fields.append((n.replace("pub", "").strip(), ty.strip()))
        if len(fields) == 1 and fields[0][0].startswith("_"):
            opaque.append(m.group(1))
        else:
            structs[m.group(1)] = [(n, rust_type(ty)) for n, ty in fields]
    return funcs, structs, opaque
