#!/bin/bash
# round 5: what device-side TxPlan hashing could give AT MOST -- the host's hashing skipped altogether (measurement hook)
O=gpurun_out/r05x; mkdir -p $O
for i in 1 2 3 4; do
  timeout 300 python3 tools/tx_inflight.py 1024 8 64 | tail -1 >> $O/inflight_real.txt 2>> $O/err.txt
  ZKGPU_TEST_TX_FREE_HASHING=1 timeout 300 python3 tools/tx_inflight.py 1024 8 64 | tail -1 >> $O/inflight_free.txt 2>> $O/err.txt
done
for i in 1 2; do
  timeout 300 python3 tools/tx_inflight.py 4096 4 32 | tail -1 >> $O/inflight4096_real.txt 2>> $O/err.txt
  ZKGPU_TEST_TX_FREE_HASHING=1 timeout 300 python3 tools/tx_inflight.py 4096 4 32 | tail -1 >> $O/inflight4096_free.txt 2>> $O/err.txt
done
timeout 300 python3 tools/tx_bench.py 8 2>&1 | grep "library call alone" >> $O/call8192_real.txt
ZKGPU_TEST_TX_FREE_HASHING=1 timeout 300 python3 tools/tx_bench.py 8 2>&1 | grep "library call alone" >> $O/call8192_free.txt
timeout 300 python3 tools/tx_bench.py 1 2>&1 | grep "library call alone" >> $O/call1024_real.txt
ZKGPU_TEST_TX_FREE_HASHING=1 timeout 300 python3 tools/tx_bench.py 1 2>&1 | grep "library call alone" >> $O/call1024_free.txt
for f in $O/*.txt; do echo == $f; cat $f; done
