#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for c in 0 1024 1536 2048; do echo "tx chunk $c"; python tools/tx_bench.py 8 0 $c 2>&1 | grep "library call" | tail -3; done
echo "16 threads explicit"; python tools/tx_bench.py 8 0 0 16 2>&1 | grep "library call" | tail -3
echo "12 threads explicit"; python tools/tx_bench.py 8 0 0 12 2>&1 | grep "library call" | tail -3
echo "24 threads explicit"; python tools/tx_bench.py 8 0 0 24 2>&1 | grep "library call" | tail -3
echo 1024; python tools/tx_bench.py 1 2>&1 | grep "library call" | tail -3
echo 16384; python tools/tx_bench.py 16 2>&1 | grep "library call" | tail -3
ZKGPU_PROVER_TIMING=1 python tools/tx_bench.py 8 > gpurun_out/tx19.log 2>&1
awk '/tx verify/{c++} c==4' gpurun_out/tx19.log | head -32
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/txtrace19 -- python3 $R/tools/tx_bench.py 8 > /dev/null 2>&1
cd $R
python tools/tx_trace.py $(ls gpurun_out/txtrace19/*/*kernel_trace.csv | head -1) 12 > gpurun_out/txtrace19.txt; head -3 gpurun_out/txtrace19.txt
