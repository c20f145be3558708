#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "zkvm_tx or never_run_over" 2>&1 | tail -8
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 2>&1 | grep -E "library call|tx verify" | tail -4
python tools/tx_bench.py 1 2>&1 | grep -E "library call" | tail -2
python tools/tx_bench.py 16 2>&1 | grep -E "library call" | tail -2
python tools/tx_bench.py 32 2>&1 | grep -E "library call" | tail -2
