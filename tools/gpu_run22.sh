#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03b_gpu_tests.log | tail -3
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r03b_smoke.log 2>&1; tail -2 gpurun_out/r03b_smoke.log
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03b_bench_driverflags.json 2> gpurun_out/r03b_bench_driverflags.err; echo "driver command wall: $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r03b_bench_driverflags.json").readline())
print("value", d["value"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"), "hbm", d.get("hbm_copy",{}).get("measured_copy_GBps"))
r=d["roofline"]; print("roofline", r["kernel"], r["frac"], r["traffic"], "step", r["step"])
m=d.get("msm_2p20",{}); print("msm", m.get("pairs_per_s"), m.get("ms"), m.get("kernel_ms_sum"), m.get("roofline",{}).get("kernel"), m.get("valu_issue_frac"))
print("tx", d.get("tx_verify")); print("sweep", d["setup"].get("table_bits_sweep"))
print("cpu", d.get("cpu_baseline",{}).get("value"), "prover", d.get("prover",{}).get("proofs_per_s"), d.get("prover_1024_constraints",{}).get("proofs_per_s"))
PY
tail -3 gpurun_out/r03b_bench_driverflags.err
bash tools/profile_bench.sh r03b > gpurun_out/r03b_profile.log 2>&1; tail -5 gpurun_out/r03b_profile.log
