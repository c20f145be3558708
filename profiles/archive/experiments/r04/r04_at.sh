#!/bin/bash
# locate-mode 3 (the locating sums of ALL groups formed beside the group sums: the culprit named two launches earlier) on the 20-step run
cd "$(dirname "$0")/../../.."
for rep in 1 2; do
  for lm in 0 3; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm --locate-mode $lm > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('locate-mode $lm rep $rep: value %.0f steady %.0f latency %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms']))"
  done
done
