#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04d}
cd $R
timeout 1500 python -m pytest tests/test_zkvm_tx.py -m gpu -x -q > gpurun_out/${TAG}_tx_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tx_tests.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc $?"; tail -3 gpurun_out/${TAG}_bench.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").readline())
print("value", d["value"])
print("tx", {k:v for k,v in d.get("tx_verify",{}).items() if k!="note"})
PY
