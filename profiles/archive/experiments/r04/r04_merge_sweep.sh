#!/bin/bash
# the driver's 20-step run by merge target (how many device batches the 20 tickets become)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for M in 10240 5120 7168 4096 6144 10240; do
  for rep in 1 2; do
    python3 bench.py --lean --steps 20 --warmup 5 --merge $M 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('merge $M', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r04_merge_sweep.txt 2>&1
cat gpurun_out/r04_merge_sweep.txt
