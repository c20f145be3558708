#!/bin/bash
cd $GRAFT_REPO_ROOT
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full.json 2> gpurun_out/full.err
echo "driver command wall: $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/full.json").readline())
print("value", d["value"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"), "hbm", d.get("hbm_copy",{}).get("measured_copy_GBps"))
r=d["roofline"]; print("roofline", r["kernel"], r["frac"], r["traffic"], "step", r["step"])
m=d.get("msm_2p20",{}); print("msm", m.get("pairs_per_s"), m.get("ms"), m.get("kernel_ms_sum"), m.get("roofline",{}).get("kernel"), m.get("valu_issue_frac"))
print("tx", d.get("tx_verify")); print("sweep", d["setup"].get("table_bits_sweep"))
print("cpu", d.get("cpu_baseline",{}).get("value"), "prover", d.get("prover",{}).get("proofs_per_s"), d.get("prover_1024_constraints",{}).get("proofs_per_s"))
print("config", {k:v for k,v in d["config"].items() if k!="workload"})
PY
tail -3 gpurun_out/full.err
S=$(date +%s); python bench.py > gpurun_out/default.json 2> gpurun_out/default.err; echo "default command wall: $(( $(date +%s) - S )) s"; python -c "import json; d=json.loads(open('gpurun_out/default.json').readline()); print(d['value'], d['steps'])"
