#!/bin/bash
cd $GRAFT_REPO_ROOT
L=zkvm_amd/lib
cp $L/libzkgpu_segments.so $L/libzkgpu.so
timeout 900 python -m pytest tests/test_zkvm_tx.py -m gpu -x -q > gpurun_out/t27.log 2>&1; grep -E "passed|failed|error" gpurun_out/t27.log | tail -3
for round in 1 2; do
for v in perchunk segments; do
  cp $L/libzkgpu_$v.so $L/libzkgpu.so
  for n in 8 16 32 64 1; do echo "$v copies $n: $(python tools/tx_bench.py $n 2>&1 | grep 'library call' | awk '{print $4}' | tr '\n' ' ')"; done
done
done
