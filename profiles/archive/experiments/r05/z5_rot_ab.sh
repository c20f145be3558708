#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/ab_rot.txt
for i in 1 2 3 4; do
  for V in rot norot; do
    L=""; [ $V = norot ] && L=build/ab/norot/libzkgpu.so
    ZKGPU_LIB=$L timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$V', d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'])" >> $O/ab_rot.txt
  done
done
cat $O/ab_rot.txt
