#!/bin/bash
# merge target of the ticket path again, now that k_prepare is 20 % shorter
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/merge_sweep.txt
for R in 1 2; do
for M in 8192 9216 10240 11264 12288; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --merge $M --no-sweep --no-cpu --no-msm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('merge $M', d['value'], d['steady_state']['tx_per_s'])" >> $O/merge_sweep.txt
done; done
cat $O/merge_sweep.txt
