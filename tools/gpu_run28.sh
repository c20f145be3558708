#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 0 1664 1152 2048 896; do
  if [ $b = 0 ]; then unset ZKGPU_TX_BATCH; else export ZKGPU_TX_BATCH=$b; fi
  for n in 8 32; do echo "batch $b copies $n: $(python tools/tx_bench.py $n 2>&1 | grep 'library call' | awk '{print $4}' | tr '\n' ' ')"; done
done
