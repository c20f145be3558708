#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 > gpurun_out/tx20.log 2>&1
grep "staging thread, chunk" gpurun_out/tx20.log | tail -9
