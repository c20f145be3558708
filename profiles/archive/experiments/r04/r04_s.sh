#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for D in 0 8192 12288; do
  echo "== ZKGPU_TX_DEFER=$D, 32768 per call"; ZKGPU_TX_DEFER=$D python3 tools/tx_call_profile.py 32768 2>&1 | tail -4
done > gpurun_out/r04s_tx_defer.txt 2>&1
cat gpurun_out/r04s_tx_defer.txt
