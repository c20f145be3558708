#!/bin/bash
cd "$(dirname "$0")/../../.."
for c in 0 2048 4096; do
  for rep in 1 2; do
    echo -n "8192 per call, chunk=$c rep=$rep: "
    TXCHUNK=$c python3 tools/tx_call_profile.py 8192 2>&1 | grep "^call" | sed 's/call \([0-9]\): \([0-9.]*\) ms.*/\2/' | tr '\n' ' '; echo
  done
done
ZKGPU_PROVER_TIMING=1 python3 tools/tx_call_profile.py 8192 > gpurun_out/r04aa_timing.txt 2>&1
