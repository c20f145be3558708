#!/bin/bash
# k_locate_fused: culprit named by suffix / prefix scans instead of two double-and-add chains; 256 threads per group
cd "$(dirname "$0")/../../.."
python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|error" /tmp/t.log | tail -1
python3 bench.py --gpus 1 --steps 6 --solo 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:v for k,v in d['solo_kernel_ms'].items() if k in ('k_locate_fused','k_recheck_fused','k_group_combine','k_msm_finish_quad')})"
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('rep $rep: value %.0f steady %.0f latency %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms']))"
done
