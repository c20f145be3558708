#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 bench.py --config 4 > gpurun_out/r04l_config4.json 2> gpurun_out/r04l_config4.err; echo "config 4 rc $?"; tail -2 gpurun_out/r04l_config4.err
ZKGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --config 4 --steps 10 --warmup 2 > gpurun_out/r04l_config4_n2.json 2> gpurun_out/r04l_config4_n2.err; echo "config 4 N=2 shared rc $?"; tail -2 gpurun_out/r04l_config4_n2.err
python3 - <<PY
import json
for f in ("r04l_config4.json","r04l_config4_n2.json"):
    d=json.loads(open("gpurun_out/"+f).readline())
    print(f, d["value"], d["n_gpus"], d["config"].get("generator_table_bits"), d["setup"]["table_bytes"], d["config"].get("rccl"), d["config"].get("exchange"))
PY
