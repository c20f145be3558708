#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 > gpurun_out/tx18.log 2>&1
grep -n "tx verify" gpurun_out/tx18.log | head -3
awk '/tx verify/{c++} c==3' gpurun_out/tx18.log | head -40
