#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --lean "$@" 2>gpurun_out/r24.err | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('one_lane=$ZKGPU_HORNER_ONE_LANE $*', '->', round(d['value']/1e6,3))" || tail -3 gpurun_out/r24.err; }
for rep in 1 2; do
unset ZKGPU_HORNER_ONE_LANE
run --steps 200 --warmup 10; run --steps 20 --warmup 5
export ZKGPU_HORNER_ONE_LANE=1
run --steps 200 --warmup 10; run --steps 20 --warmup 5; run --steps 200 --warmup 10 --inflight 7
done
