#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r04j_gpu_tests.log 2>&1; tail -3 gpurun_out/r04j_gpu_tests.log
for rep in 1 2 3; do
  python3 bench.py --lean --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('two-phase dispatch', d['value'], d['ms_per_step'])"
done > gpurun_out/r04j_twophase.txt 2>&1
python3 bench.py --lean --steps 200 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('200 steps', d['value'], d['ms_per_step'])" >> gpurun_out/r04j_twophase.txt 2>&1
cat gpurun_out/r04j_twophase.txt
