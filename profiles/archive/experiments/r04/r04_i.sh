#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for E in 0 1; do
    ZKGPU_BENCH_NO_INFLIGHT_EVENTS=$E python3 bench.py --lean --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('no_events=$E', d['value'], d['ms_per_step'], d.get('host_submit_ms_per_step'))"
  done
done > gpurun_out/r04i_events_cost.txt 2>&1
cat gpurun_out/r04i_events_cost.txt
