#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04k -- python3 $R/bench.py --lean --steps 20 --warmup 5 > $R/gpurun_out/prof_r04k_bench.json 2> $R/gpurun_out/prof_r04k.err
python3 $R/tools/trace_region.py $R/gpurun_out/prof_r04k/*/*_kernel_trace.csv > $R/gpurun_out/r04k_timeline.txt 2>&1
head -60 $R/gpurun_out/r04k_timeline.txt
