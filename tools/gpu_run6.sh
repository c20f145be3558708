#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/txtr -- python3 $R/tools/tx_bench.py 8 > /dev/null 2> $R/gpurun_out/txtr.err)
python tools/tx_trace.py $(ls gpurun_out/txtr/*/*_kernel_trace.csv | head -1) 14 > gpurun_out/txtr_summary.txt
grep -v "fillBuffer\|copyBuffer" gpurun_out/txtr_summary.txt | awk '$2 > 0.05 || /last burst/' | head -90
