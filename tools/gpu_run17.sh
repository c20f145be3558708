#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --lean "$@" 2>gpurun_out/r17.err | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('reserve=$ZKGPU_RESERVE_CUS $*', '->', round(d['value']/1e6,3))" || tail -3 gpurun_out/r17.err; }
for r in 0 8 16 32 64; do
  export ZKGPU_RESERVE_CUS=$r
  run --steps 20 --warmup 5; run --steps 20 --warmup 5; run --steps 200 --warmup 10
done
