#!/bin/bash
cd $GRAFT_REPO_ROOT
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 0 0 16 2>&1 | grep -E "library call|tx verify" | tail -2
ZKGPU_TX_TAIL_SPLIT=1 ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 0 0 16 2>&1 | grep -E "library call|tx verify" | tail -2
for TC in 1536 2048 4096; do ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 0 $TC 16 2>&1 | grep -E "library call" | tail -1; done
