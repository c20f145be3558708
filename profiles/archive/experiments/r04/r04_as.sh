#!/bin/bash
# chip-filling streams with a CU mask that leaves every k-th CU to the light streams (ZKGPU_RESERVE_CUS=k)
cd "$(dirname "$0")/../../.."
for k in 0 1 2 4; do
  for q in 18; do
    ZKGPU_RESERVE_CUS=$k GPU_MAX_HW_QUEUES=$q python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); t=d['tx_verify']
print('reserve=$k q=$q: value %.0f steady %.0f latency %s host %s | tx 8192 %s 32768 %s inflight %s | msm %.0f prover %.0f lanes %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s'], t['ms_8192_per_call'], t['ms_32768_per_call'], t['in_flight']['tx_per_s'], d['msm_2p20']['pairs_per_s'], d['prover']['proofs_per_s'], t.get('lanes')))" 2>&1 | tail -1
    
  done
done
