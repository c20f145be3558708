#!/bin/bash
# k_small_accumulate at 128 registers (four wavefronts per SIMD, two registers spilled) against 129 (three)
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/small_occ4.txt
for R in 1 2 3; do
for V in base occ4; do
  L=""; [ $V = occ4 ] && L=build/ab/occ4/libzkgpu.so
  ZKGPU_LIB=$L timeout 600 python3 bench.py --solo --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('solo $V', round(d['solo_kernel_ms']['k_small_accumulate'],4))" >> $O/small_occ4.txt
  ZKGPU_LIB=$L timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('bench $V', d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'])" >> $O/small_occ4.txt
done; done
cat $O/small_occ4.txt
