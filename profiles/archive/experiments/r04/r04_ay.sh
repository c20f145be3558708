#!/bin/bash
# three transaction rounds in flight (ZKGPU_TX_ROUNDS=3) against two
cd "$(dirname "$0")/../../.."
for rep in 1 2 3; do
  for r in 2 3; do
    echo -n "rounds=$r rep=$rep: "
    ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 1024 8 64 2>&1 | tail -2 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s; \(.*\)/\2 (\3)/' | tr '\n' ' '
    echo -n " | 1024x12: "
    ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 1024 12 72 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s; \(.*\)/\2 (\3)/' | tr '\n' ' '
    echo -n " | 4096x4: "
    ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 4096 4 32 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s.*/\2/' | tr '\n' ' '
    echo
  done
done
