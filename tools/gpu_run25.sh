#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_zkvm_tx.py -m gpu -x -q > gpurun_out/t25.log 2>&1; grep -E "passed|failed|error" gpurun_out/t25.log | tail -3
for n in 8 8 1 4 16 32; do echo "copies $n"; python tools/tx_bench.py $n 2>&1 | grep "library call" | tail -3; done
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 > gpurun_out/tx25.log 2>&1
awk '/tx verify/{c++} c==5' gpurun_out/tx25.log | head -30
