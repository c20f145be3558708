#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "zkvm_tx or never_run_over" 2>&1 | tail -2
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 0 0 16 2>&1 | grep -v "^txblock" | grep -B22 "library call alone" | tail -24
python tools/tx_bench.py 16 0 0 16 2>&1 | grep "library call alone" | tail -2
python tools/tx_bench.py 1 0 0 16 2>&1 | grep "library call alone" | tail -2
