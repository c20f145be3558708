#!/bin/bash
# round 4, second GPU call: the new host-memory ticket test + small-row test, then the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04b}
cd $R
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_msm.py -m gpu -x -q -k "host_memory or small_rows or benched or exchange_step" > gpurun_out/${TAG}_new_tests.log 2>&1; tail -5 gpurun_out/${TAG}_new_tests.log
S=$(date +%s); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-msm --no-cpu > gpurun_out/${TAG}_bench_driverflags.json 2> gpurun_out/${TAG}_bench_driverflags.err; echo "driver command: rc $? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/${TAG}_bench_driverflags.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_driverflags.json").readline())
print("value", d["value"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"))
print("hostmem", {k:v for k,v in d.get("host_memory",{}).items() if k!="note"})
PY
