#!/bin/bash
# two transaction rounds in flight again, by hardware queues (the r04e / r04f measurements were taken at 24)
cd "$(dirname "$0")/../../.."
for q in 12 14 16 18; do
  for rep in 1 2 3; do
    for r in 1 2; do
      echo -n "q=$q rounds=$r rep=$rep: "
      GPU_MAX_HW_QUEUES=$q ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 1024 8 64 2>&1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s.*/\2/' | tr '\n' ' '
      echo -n " | 4096x4: "
      GPU_MAX_HW_QUEUES=$q ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 4096 4 32 2>&1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s.*/\2/' | tr '\n' ' '
      echo
    done
  done
done
