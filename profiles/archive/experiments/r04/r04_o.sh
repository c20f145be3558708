#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r04o_gpu_tests.log 2>&1; tail -3 gpurun_out/r04o_gpu_tests.log
for BIF in 1 2 3 4; do
  for rep in 1 2; do
    python3 bench.py --config 4 --lean --blocks-in-flight $BIF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config 4, blocks in flight $BIF:', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r04o_config4_blocks_in_flight.txt 2>&1
cat gpurun_out/r04o_config4_blocks_in_flight.txt
python3 bench.py --lean --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config 2, 20 steps:', d['value'])"
