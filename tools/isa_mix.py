"""The integer roofline that binds ("report HBM fraction as mandated + integer-multiply
rate; say which binds"): per kernel, the histogram of its VALU opcodes from the compiler's assembly, weighted by loop depth,
times the measured issue cost of every opcode (tools/ubench/valu_ops.hip on an MI355X -> profiles/r05_valu_op_rates.txt)
= the kernel's MIX-WEIGHTED cycles per wave-instruction.  A launch that retires N vector wave-instructions (SQ_INSTS_VALU,
profiles/pmc_valu*.json) cannot finish before   N x cpi_mix / (SIMDs x clock)   -- its `mix_bound_ms`; bench.py divides
that by the measured duration (`mix_frac`) and reports the multiply-adds per second beside their own peak.

  python3 tools/isa_mix.py [--asm /tmp/zkgpu.s] [--rates profiles/r05_valu_op_rates.txt] [--out profiles/valu_mix.json]

Without --asm the library's device code is compiled to assembly first (hipcc -S --cuda-device-only: ~3 minutes).
The mix is taken over the kernel's HOT loops -- the loops (a backward branch to an earlier label closes one) whose bodies
hold at least half as many vector instructions as the largest: the kernels here are a prologue, one or two loops of fully
unrolled field arithmetic, an epilogue -- or over the whole body when there is no such loop; the static histogram is kept
beside it.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# opcodes the micro-benchmark does not time, priced as a timed relative of the same unit and width
ALIAS = {
    "v_subrev_u32": "v_sub_u32", "v_sub_co_u32": "v_add_co_u32", "v_subrev_co_u32": "v_add_co_u32", "v_subbrev_co_u32": "v_subb_co_u32",
    "v_add_i32": "v_add_u32", "v_sub_i32": "v_sub_u32", "v_not_b32": "v_xor_b32", "v_xnor_b32": "v_xor_b32", "v_or3_b32": "v_add3_u32",
    "v_xad_u32": "v_add3_u32", "v_xor3_b32": "v_add3_u32", "v_ashrrev_i32": "v_lshrrev_b32", "v_ashrrev_i64": "v_lshrrev_b64", "v_bfe_i32": "v_bfe_u32",
    "v_min_u32": "v_add_u32", "v_max_u32": "v_add_u32", "v_min_i32": "v_add_u32", "v_max_i32": "v_add_u32", "v_mul_i32_i24": "v_mul_u32_u24",
    "v_mad_i32_i24": "v_mad_u32_u24", "v_mul_hi_i32": "v_mul_hi_u32", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32",
    "v_accvgpr_mov_b32": "v_mov_b32", "v_lshlrev_b16": "v_lshlrev_b32", "v_perm_b32": "v_bfi_b32", "v_alignbyte_b32": "v_alignbit_b32",
    "v_cmp": "v_cmp_lt_u32", "v_cmpx": "v_cmp_lt_u32", "v_writelane_b32": "v_readlane_b32", "v_permlane16_swap_b32": "v_mov_b32_dpp_row_ror",
    "v_permlane32_swap_b32": "v_mov_b32_dpp_row_ror", "v_bcnt_u32_b32": "v_add_u32", "v_mbcnt_lo_u32_b32": "v_add_u32", "v_mbcnt_hi_u32_b32": "v_add_u32",
    "v_ffbh_u32": "v_add_u32", "v_ffbl_b32": "v_add_u32", "v_lshl_add_u64": "v_lshl_add_u64", "v_pk_mov_b32": "v_pk_mov_b32",
    "v_add_nc_u32": "v_add_u32", "v_sad_u32": "v_add3_u32", "v_cvt_f32_u32": "v_add_u32", "v_cvt_u32_f32": "v_add_u32", "v_rcp_iflag_f32": "v_mul_lo_u32",
    "v_mul_f32": "v_add_u32", "v_fma_f32": "v_add_u32", "v_mac_f32": "v_add_u32", "v_fmac_f32": "v_add_u32", "v_add_f32": "v_add_u32", "v_trunc_f32": "v_add_u32",
    "v_swap_b32": "v_mov_b32", "v_nop": None,
}
MADS = ("v_mad_u64_u32", "v_mad_i64_i32")


def load_rates(path):
    rates = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        name, cyc = line.split()[:2]
        rates[name] = float(cyc)
    return rates


def kernels_of(asm_text):
    """-> {short kernel name: [lines]} for every .amdhsa kernel and (non-inlined) device function of namespace zk"""
    out, name, body = {}, None, []
    for line in asm_text.splitlines():
        m = re.match(r"^(_ZN2zk\w+):\s*(;.*)?$", line)
        if m:
            if name:
                out.setdefault(name, []).extend(body)
            sym = m.group(1)
            d = re.match(r"_ZN2zk(\d+)", sym)
            short = sym[len(d.group(0)): len(d.group(0)) + int(d.group(1))] if d else sym
            name, body = short, []
            continue
        if line.startswith(".Lfunc_end") and name:
            out.setdefault(name, []).extend(body)
            name, body = None, []
            continue
        if name is not None:
            body.append(line)
    return out


def opcode_key(op, text):
    if op.startswith("v_cmpx_"):
        return "v_cmpx"
    if op.startswith("v_cmp_"):
        return "v_cmp"
    if op.endswith("_dpp"):
        base = op[:-4]
        if "quad_perm" in text:
            return "v_mov_b32_dpp_quad" if base == "v_mov_b32" else "v_add_u32_dpp_quad"
        return "v_mov_b32_dpp_row_ror"
    if op.startswith("v_cndmask_b32"):          # the mask in VCC costs more than in an SGPR pair (profiles/r05_valu_op_rates.txt)
        return "v_cndmask_b32_e32_vcc" if (op.endswith("_e32") or text.rstrip().endswith("vcc")) else "v_cndmask_b32_e64_sgpr"
    for suffix in ("_e32", "_e64", "_sdwa"):
        if op.endswith(suffix):
            op = op[: -len(suffix)]
    return op


def analyse(lines, rates):
    labels, insts = {}, []           # label -> instruction index; (opcode key, is branch target label or None)
    for line in lines:
        s = line.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            t = s.split()[-1]
            tgt = t if t.startswith(".LBB") else None
        insts.append((op, s, tgt))
    # loops: a backward branch to an earlier label closes [label, branch].  HOT = the loops whose bodies hold at least half
    # as many vector instructions as the largest one (the kernels here are a prologue, one or two loops of fully unrolled
    # field arithmetic, an epilogue; small inner loops -- a skip-ahead, a digit scan -- are control flow, not the mix)
    loops = []
    for i, (op, s, tgt) in enumerate(insts):
        if tgt and tgt in labels and labels[tgt] <= i:
            lo = labels[tgt]
            loops.append((sum(1 for q in insts[lo: i + 1] if q[0].startswith("v_")), lo, i))
    hot = set()
    if loops:
        big = max(l[0] for l in loops)
        for n, lo, hi in loops:
            if n * 2 >= big and n >= 100:
                hot.update(range(lo, hi + 1))
    static, weighted, unknown = collections.Counter(), collections.Counter(), collections.Counter()
    n_valu = n_salu = n_mem = 0
    for i, (op, s, _) in enumerate(insts):
        if op.startswith("v_"):
            key = opcode_key(op, s)
            key = ALIAS.get(key, key)
            if key is None:
                continue
            if key not in rates:
                unknown[key] += 1
                key = "v_add_u32"
            static[key] += 1
            if not hot or i in hot:
                weighted[key] += 1
            n_valu += 1
        elif op.startswith("s_"):
            n_salu += 1
        else:
            n_mem += 1
    tot = sum(weighted.values())
    if not tot:
        return None
    cpi = sum(rates[k] * v for k, v in weighted.items()) / tot
    cpi_static = sum(rates[k] * v for k, v in static.items()) / max(1, sum(static.values()))
    mad = sum(weighted[k] for k in MADS) / tot
    top = {k: round(v / tot, 4) for k, v in weighted.most_common(12)}
    return {"cpi_mix": round(cpi, 3), "cpi_static": round(cpi_static, 3), "mad_frac": round(mad, 4), "valu_static": n_valu, "salu_static": n_salu,
            "other_static": n_mem, "hot_loop_valu": int(tot) if hot else 0, "mix": top,
            "unpriced_as_v_add_u32": dict(unknown) or None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None)
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r05_valu_op_rates.txt"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_mix.json"))
    ap.add_argument("--show", nargs="*", default=[])
    args = ap.parse_args()
    asm = args.asm
    if not asm:
        asm = "/tmp/zkgpu_isa_mix.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "zkvm_amd", "csrc", "zkgpu.hip"), "-o", asm], check=True)
    rates = load_rates(args.rates)
    ks = kernels_of(open(asm).read())
    table = {"_note": "tools/isa_mix.py: per kernel (and non-inlined device function), VALU opcodes of the compiler's assembly over its hot loops "
                      "x the issue cost of each opcode (%s) -> cpi_mix = cycles one SIMD spends per vector wave-instruction of this kernel; "
                      "mad_frac = share of v_mad_u64_u32 among them.  A kernel that CALLS non-inlined functions (the prover's phases) shows "
                      "only its own body." % os.path.relpath(args.rates, ROOT),
             "_rates": rates}
    for name, lines in sorted(ks.items()):
        r = analyse(lines, rates)
        if r and r["valu_static"] >= 20:
            table[name] = r
    json.dump(table, open(args.out, "w"), indent=1, sort_keys=True)
    for name in args.show or [k for k in table if k.startswith("k_")]:
        if name in table and not name.startswith("_"):
            r = table[name]
            print("%-28s cpi_mix %.2f (static %.2f)  mad %.0f%%  valu %6d  hot %5d  %s" % (name, r["cpi_mix"], r["cpi_static"], 100 * r["mad_frac"],
                  r["valu_static"], r["hot_loop_valu"], " ".join("%s:%.0f%%" % (k[2:], 100 * v) for k, v in list(r["mix"].items())[:7])))


if __name__ == "__main__":
    main()
