#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for M in 4096 8192; do
for BIF in 1 2 3 4; do
  for rep in 1 2; do
    python3 bench.py --config 4 --lean --blocks-in-flight $BIF --merge $M 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config 4, merge $M, blocks in flight $BIF:', d['value'], d['ms_per_step'])"
  done
done
done > gpurun_out/r04p_config4_blocks_in_flight.txt 2>&1
cat gpurun_out/r04p_config4_blocks_in_flight.txt
