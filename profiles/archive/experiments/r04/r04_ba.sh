#!/bin/bash
# the driver's command ten times in a row on one box: the spread of `value` (20 timed steps = 7 ms) and of the steady state
cd "$(dirname "$0")/../../.."
for rep in 1 2 3 4 5 6 7 8 9 10; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('run $rep: value %.0f steady %.0f latency %s host %s inflight %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s'], d['tx_verify']['in_flight']['tx_per_s'] if 'tx_verify' in d else None))"
done
