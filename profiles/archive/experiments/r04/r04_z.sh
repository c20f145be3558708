#!/bin/bash
cd "$(dirname "$0")/../../.."
for rep in 1 2 3; do
  echo "default rounds, rep=$rep"; python3 tools/tx_inflight.py 1024 8 64 2>&1 | tail -2
  python3 tools/tx_inflight.py 4096 4 32 2>&1 | tail -2
  python3 tools/tx_inflight.py 2048 8 32 2>&1 | tail -1
done

