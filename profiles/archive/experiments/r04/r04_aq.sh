#!/bin/bash
# the three-piece dispatch + the aligned-word merge kernel as kept: GPU tests, the driver's run x3 (full legs once), config 4 x2
cd "$(dirname "$0")/../../.."
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); t=d['tx_verify']
print('rep $rep: value %.0f steady %.0f latency %s host %s | tx 8192 %s 32768 %s inflight %s | msm %.0f prover %.0f' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s'], t['ms_8192_per_call'], t['ms_32768_per_call'], t['in_flight']['tx_per_s'], d['msm_2p20']['pairs_per_s'], d['prover']['proofs_per_s']))"
done
for rep in 1 2; do python3 bench.py --config 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config4', d['value'])"; done
