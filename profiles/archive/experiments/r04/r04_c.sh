#!/bin/bash
# round 4, third GPU call: the transaction path after the TxCall refactor (+ calls in flight), the second-verifier warning,
# the table-width knee, then the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04c}
cd $R
timeout 1500 python -m pytest tests/test_zkvm_tx.py tests/test_gpu_block.py tests/test_gpu_msm.py -m gpu -x -q -k "transaction or second_verifier or table_width or synchronous or lanes_are_probed or long_call or fixture" > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log
S=$(date +%s); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driverflags.json 2> gpurun_out/${TAG}_bench_driverflags.err; echo "driver command: rc $? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/${TAG}_bench_driverflags.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_driverflags.json").readline())
print("value", d["value"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"), "table bits", d["config"]["generator_table_bits"], d["setup"]["table_bytes"])
print("hostmem", {k:v for k,v in d.get("host_memory",{}).items() if k!="note"})
print("tx", {k:v for k,v in d.get("tx_verify",{}).items() if k!="note"})
print("hbm", d.get("hbm_copy",{}).get("measured_copy_GBps"), "msm", d.get("msm_2p20",{}).get("pairs_per_s"), "sweep", {k:v.get("tx_per_s") for k,v in d["setup"].get("table_bits_sweep",{}).items() if isinstance(v,dict)})
PY
