#!/bin/bash
cd $GRAFT_REPO_ROOT
g++ -O2 -std=c++17 -pthread -I zkvm_amd/csrc -o /tmp/vm_threads tools/ubench/vm_threads.cpp && /tmp/vm_threads 2>&1 | grep -A1 -E "^ ?(4|8|16|32) threads"
for n in 8 8 1 16 32; do echo "copies $n"; python tools/tx_bench.py $n 2>&1 | grep "library call" | tail -3; done
