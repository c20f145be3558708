#!/bin/bash
# generator-table rows at a 128-byte stride (one row per line) instead of 96: parity tests, bench, config 4, prover
cd "$(dirname "$0")/../../.."
python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q -k "table or tables or oracle or prover or verif" > /tmp/t.log 2>&1; grep -E "passed|failed|error" /tmp/t.log | tail -1
for rep in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); t=d['tx_verify']
print('rep $rep: value %.0f steady %.0f latency %s host %s | tx 8192 %s | prover %.0f %.0f | table bytes %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['host_memory']['tickets']['tx_per_s'], t['ms_8192_per_call'], d['prover']['proofs_per_s'], d['prover_1024_constraints']['proofs_per_s'], d['setup'].get('table_bytes')))"
done
python3 bench.py --config 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config4', d['value'])"
