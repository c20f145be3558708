#!/bin/bash
# the serialized-transaction call at 32768 (and 8192) per call by GPU_MAX_HW_QUEUES, several processes each: which mode does a process land in?
cd "$(dirname "$0")/../../.."
for q in 8 12 16 20 24; do
  for rep in 1 2 3 4; do
    for n in 32768; do
      echo -n "q=$q n=$n rep=$rep: "
      GPU_MAX_HW_QUEUES=$q python3 tools/tx_call_profile.py $n 2>&1 | grep "^call\|Error\|error" | sed 's/call \([0-9]\): \([0-9.]*\) ms.*/\2/' | tr '\n' ' '
      echo
    done
  done
done
