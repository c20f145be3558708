#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
timeout 600 python -m pytest tests/test_zkvm_tx.py -m gpu -x -q -k "in_flight or long_call" 2>&1 | tail -2
for A in "1024 8 64" "1024 16 96" "2048 4 32" "4096 4 16"; do
  echo "== two rounds: $A"; python3 tools/tx_inflight.py $A 2>&1 | tail -2
  echo "== one round: $A"; ZKGPU_TX_ROUNDS=1 python3 tools/tx_inflight.py $A 2>&1 | tail -2
done
} > gpurun_out/r04e_inflight.txt 2>&1
cat gpurun_out/r04e_inflight.txt
