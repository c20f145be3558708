#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "== timing: one round, 2048 4 8"; ZKGPU_TX_ROUNDS=1 ZKGPU_PROVER_TIMING=1 python3 tools/tx_inflight.py 2048 4 8 2>&1 | tail -150
} > gpurun_out/r04f_inflight_timing.txt 2>&1
tail -120 gpurun_out/r04f_inflight_timing.txt
