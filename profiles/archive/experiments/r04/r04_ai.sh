#!/bin/bash
# the failure path's arrangements again at the end of round 4 (18 queues): Horner chains per transaction / per group first
cd "$(dirname "$0")/../../.."
for rep in 1 2; do
  for hm in 0 1 2; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm --horner-mode $hm > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('horner-mode $hm rep $rep: value %.0f steady %.0f latency %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms']))"
  done
done
