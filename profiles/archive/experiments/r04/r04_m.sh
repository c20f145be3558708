#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for RAMP in 0 1024 2048 3072 4096 0; do
  for rep in 1 2; do
    ZKGPU_TICKET_RAMP=$RAMP python3 bench.py --lean --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ramp $RAMP', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r04m_ramp.txt 2>&1
cat gpurun_out/r04m_ramp.txt
