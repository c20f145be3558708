#!/bin/bash
# the chip-filling streams (points / accumulate / prepare) at the MIDDLE priority, the light and stage streams above them
cd "$(dirname "$0")/../../.."
python3 -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); a=ctypes.c_int(); b=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(a),ctypes.byref(b)); print('priority range least', a.value, 'greatest', b.value)"
for cfg in "high 18" "mid 18" "mid 14"; do
  set -- $cfg
  for rep in 1 2; do
    ZKGPU_HEAVY_PRIO=$1 GPU_MAX_HW_QUEUES=$2 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); t=d['tx_verify']
print('heavy=$1 q=$2 rep $rep: value %.0f steady %.0f latency %s host %s | tx 8192 %s 32768 %s inflight %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s'], t['ms_8192_per_call'], t['ms_32768_per_call'], t['in_flight']['tx_per_s']))"
  done
done
