#!/bin/bash
# the transaction path's stage contexts on ONE stream each (16 high-priority streams in the process instead of 20)
cd "$(dirname "$0")/../../.."
python3 -m pytest tests/test_zkvm_tx.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|error" /tmp/t.log | tail -1
for rep in 1 2 3 4; do
  echo -n "rep=$rep: 1024x8 "
  python3 tools/tx_inflight.py 1024 8 64 2>&1 | tail -2 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s; \(.*\)/\2/' | tr '\n' ' '
  echo -n " | 1024x12: "
  python3 tools/tx_inflight.py 1024 12 72 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s; \(.*\)/\2/' | tr '\n' ' '
  echo -n " | 4096x4: "
  python3 tools/tx_inflight.py 4096 4 32 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\), \([0-9]*\) tx.s.*/\2/' | tr '\n' ' '
  echo -n " | call 8192: "
  python3 tools/tx_call_profile.py 8192 2>&1 | grep "^call" | sed 's/call \([0-9]\): \([0-9.]*\) ms.*/\2/' | tr '\n' ' '
  echo
done
