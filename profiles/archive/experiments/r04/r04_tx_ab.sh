#!/bin/bash
# A/B of the serialized-transaction call between two trees (library + host mirror) on ONE box
cd "$(dirname "$0")/../../.."
for rep in 1 2; do
  for tree in build/ab/r04z .; do
    for n in 8192 32768; do
      echo "== $tree n=$n rep=$rep"
      (cd $tree && python3 tools/tx_call_profile.py $n 2>&1 | grep "^call\|Error\|error" | tr '\n' ';')
      echo
    done
  done
done
