#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for M in 10240 20480 16384; do
  for S in 20 200; do
    python3 bench.py --lean --steps $S --warmup 5 --merge $M 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('merge $M steps $S', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r04n_merge_big.txt 2>&1
cat gpurun_out/r04n_merge_big.txt
