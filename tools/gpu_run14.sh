#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
nproc; lscpu | grep -E "Model name|Thread|Core|Socket|MHz" | head; cat /sys/fs/cgroup/cpu.max 2>/dev/null
g++ -O2 -std=c++17 -pthread -I zkvm_amd/csrc -o /tmp/vm_threads tools/ubench/vm_threads.cpp && /tmp/vm_threads 2>&1 | grep -v empty
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 2>&1 | tail -60
