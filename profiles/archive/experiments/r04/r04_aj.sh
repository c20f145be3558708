#!/bin/bash
# three sets of shared chip-filling streams instead of two (STREAM_SETS), by hardware queues
cd "$(dirname "$0")/../../.."
for q in 18 20; do
  for rep in 1 2; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('sets=3 q=$q rep $rep: value %.0f steady %.0f latency %s host %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s']))"
  done
done
GPU_MAX_HW_QUEUES=18 python3 bench.py --config 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config4', d['value'])"
