#!/bin/bash
# round 5: a one-chunk call's MuSig hashing beside its gather (tx_call.hpp) -- A/B on one box, with the per-stage marks.
# REJECTED (profiles/r05x_*): the overlap and its switch ZKGPU_AB_TX_NO_OVERLAP are not in the tree any more; kept as the record of the command.
O=gpurun_out/r05y; mkdir -p $O
nproc > $O/nproc.txt
for i in 1 2 3; do
  ZKGPU_AB_TX_NO_OVERLAP=1 timeout 300 python3 tools/tx_bench.py 8 2>&1 | grep "library call alone" >> $O/call8192_before.txt
  timeout 300 python3 tools/tx_bench.py 8 2>&1 | grep "library call alone" >> $O/call8192_overlap.txt
  ZKGPU_AB_TX_NO_OVERLAP=1 timeout 300 python3 tools/tx_bench.py 1 2>&1 | grep "library call alone" >> $O/call1024_before.txt
  timeout 300 python3 tools/tx_bench.py 1 2>&1 | grep "library call alone" >> $O/call1024_overlap.txt
done
ZKGPU_AB_TX_NO_OVERLAP=1 ZKGPU_PROVER_TIMING=1 timeout 300 python3 tools/tx_bench.py 8 > $O/marks8192_before.txt 2>&1
ZKGPU_PROVER_TIMING=1 timeout 300 python3 tools/tx_bench.py 8 > $O/marks8192_overlap.txt 2>&1
ZKGPU_TEST_TX_FREE_HASHING=1 ZKGPU_PROVER_TIMING=1 timeout 300 python3 tools/tx_bench.py 8 > $O/marks8192_free.txt 2>&1
for i in 1 2 3; do
  ZKGPU_AB_TX_NO_OVERLAP=1 timeout 300 python3 tools/tx_inflight.py 1024 8 64 | tail -1 >> $O/inflight_before.txt 2>> $O/err.txt
  timeout 300 python3 tools/tx_inflight.py 1024 8 64 | tail -1 >> $O/inflight_overlap.txt 2>> $O/err.txt
done
for i in 1 2; do
  ZKGPU_AB_TX_NO_OVERLAP=1 timeout 300 python3 tools/tx_inflight.py 4096 4 32 | tail -1 >> $O/inflight4096_before.txt 2>> $O/err.txt
  timeout 300 python3 tools/tx_inflight.py 4096 4 32 | tail -1 >> $O/inflight4096_overlap.txt 2>> $O/err.txt
done
for f in $O/call*.txt $O/inflight*.txt $O/nproc.txt; do echo == $f; cat $f; done
