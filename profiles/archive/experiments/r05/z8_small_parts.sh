#!/bin/bash
# k_small_accumulate: wavefronts per multiscalar multiplication (parts) against the quantisation of its launch
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/small_parts.txt
for P in 0 2 4 0 2; do
  ZKGPU_AB_SMALL_PARTS=$P timeout 600 python3 bench.py --solo --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('solo parts=$P', round(d['solo_kernel_ms']['k_small_accumulate'],4))" >> $O/small_parts.txt
  ZKGPU_AB_SMALL_PARTS=$P timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('bench parts=$P', d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'])" >> $O/small_parts.txt
done
cat $O/small_parts.txt
