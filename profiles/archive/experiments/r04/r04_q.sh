#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for M in 8192 10240 16384; do
for BIF in 4 6 8; do
    python3 bench.py --config 4 --lean --blocks-in-flight $BIF --merge $M 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config 4, merge $M, blocks in flight $BIF:', d['value'], d['ms_per_step'])"
done
done > gpurun_out/r04q_config4_sweep2.txt 2>&1
cat gpurun_out/r04q_config4_sweep2.txt
