#!/bin/bash
# round 4, first GPU call: GPU tests, the self-launched N = 2 rehearsal (both ranks on one GPU), the driver's N = 1 command
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04a}
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
S=$(date +%s); ZKGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_n2_selflaunch.json 2> gpurun_out/${TAG}_n2_selflaunch.err; echo "N=2 self-launched: rc $? wall $(( $(date +%s) - S )) s"; head -c 600 gpurun_out/${TAG}_n2_selflaunch.json; echo
S=$(date +%s); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driverflags.json 2> gpurun_out/${TAG}_bench_driverflags.err; echo "driver command: rc $? wall $(( $(date +%s) - S )) s"
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_driverflags.json").readline())
print("value", d["value"], "latency", d.get("latency_one_batch_ms"), "steady", d.get("steady_state",{}).get("tx_per_s"), "hostmem", d.get("host_memory",{}).get("gpu_resident_tx_per_s"))
print("tx", {k:v for k,v in d.get("tx_verify",{}).items() if k!="note"})
print("msm", d.get("msm_2p20",{}).get("pairs_per_s"))
PY
