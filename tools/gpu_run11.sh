#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmcy_ICACHE
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --solo --steps 6 > $D.out 2> $D.err
echo rc=$?
tail -5 $D.out; grep -v rocprofv3 $D.err | tail -20
