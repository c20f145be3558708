"""The integer roofline's plumbing ("report HBM fraction as mandated + integer-multiply rate; say which
binds"): tools/isa_mix.py on a small assembly text, the committed tables bench.py reads, and bench.valu_roofline on them.
CPU only: no kernel runs here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ASM = """
_ZN2zk6k_demoEPj:                        ; @_ZN2zk6k_demoEPj
\tv_mov_b32_e32 v1, 0
\tv_add_u32_e32 v2, v0, v1
.LBB0_1:
%s
\tv_cndmask_b32_e64 v9, v9, v8, s[4:5]
\tv_cndmask_b32_e32 v9, v9, v8, vcc
\ts_cbranch_scc1 .LBB0_1
\tv_and_b32_e32 v3, 3, v2
\ts_endpgm
.Lfunc_end0:
""" % "\n".join(["\tv_mad_u64_u32 v[4:5], s[6:7], v2, v3, v[4:5]"] * 150 + ["\tv_and_b32_e32 v6, 0x3ffffff, v4"] * 50)


def test_isa_mix_prices_the_hot_loop_of_a_kernel():
    import isa_mix
    rates = isa_mix.load_rates(os.path.join(ROOT, "profiles", "r05_valu_op_rates.txt"))
    assert 4.0 < rates["v_mad_u64_u32"] < 4.6 and 1.9 < rates["v_and_b32"] < 2.6 and rates["v_cndmask_b32_e32_vcc"] > rates["v_cndmask_b32_e64_sgpr"]
    ks = isa_mix.kernels_of(ASM)
    assert list(ks) == ["k_demo"]
    r = isa_mix.analyse(ks["k_demo"], rates)
    assert r["valu_static"] == 205 and r["hot_loop_valu"] == 202          # the loop holds 150 + 50 + 2; prologue and epilogue are not the mix
    assert abs(r["mad_frac"] - 150 / 202) < 1e-3
    want = (150 * rates["v_mad_u64_u32"] + 50 * rates["v_and_b32"] + rates["v_cndmask_b32_e64_sgpr"] + rates["v_cndmask_b32_e32_vcc"]) / 202
    assert abs(r["cpi_mix"] - want) < 2e-3


def test_committed_tables_cover_the_headline_kernels_and_bench_prices_them():
    import bench
    mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    valu = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
    for k in ("k_small_accumulate", "k_points_tables", "k_prepare", "k_static_accumulate", "k_msm_finish_quad", "k_bucket_accumulate", "k_pow22523"):
        assert k in mix and 3.0 < mix[k]["cpi_mix"] < 5.0 and 0.3 < mix[k]["mad_frac"] < 0.7, k
    units = valu["_units_per_launch"]
    v = bench.valu_roofline("k_small_accumulate", 0.95, valu, units)
    assert v["kernel"] == "k_small_accumulate" and v["bound"] == "valu-int"
    assert v["valu_wave_instructions_per_launch"] == valu["k_small_accumulate"]
    # the bound is what the instruction stream costs at its own mix: instructions x cpi / (1024 SIMDs x 2.4 GHz)
    assert abs(v["mix_bound_ms"] - valu["k_small_accumulate"] * mix["k_small_accumulate"]["cpi_mix"] / (1024 * 2.4e9) * 1e3) < 1e-3
    assert 0.5 < v["mix_frac"] <= 1.0 and 0.3 < v["frac"] < v["mix_frac"]
    assert abs(v["peak_Gmad_s"] - 1024 * 64 * 2.4 / mix["_rates"]["v_mad_u64_u32"]) < 1.0
    # twice the transactions per launch: twice the instructions, twice the bound
    v2 = bench.valu_roofline("k_small_accumulate", 1.9, valu, 2 * units)
    assert abs(v2["mix_bound_ms"] - 2 * v["mix_bound_ms"]) < 1e-3
    assert bench.valu_roofline("no_such_kernel", 1.0, valu, units) is None


def test_bench_times_the_library_default_merge_target_not_one_of_its_own():
    """the merge target the headline runs on must be the one an integrator gets.  bench.py calls
    zkgpu_verifier_set_merge only when --merge is given; the constant it uses for its own arithmetic (ring length, priming,
    the solo pass's device batch) is the library's default, held equal here."""
    import re
    src = open(os.path.join(ROOT, "zkvm_amd", "csrc", "session.hpp")).read()
    m = re.search(r"size_t merge_target = (\d+);", src)
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    lib_merge = int(re.search(r"^LIBRARY_MERGE = (\d+)", bench_src, re.M).group(1))
    assert m and int(m.group(1)) == lib_merge == 10240
    assert re.search(r"if args\.merge_given:\s*\n\s*bv\.set_merge\(args\.merge\)", bench_src)
    assert bench_src.count("bv.set_merge(") == 2          # (the other: --config 4, its own explicit 8192)
