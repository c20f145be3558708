#!/bin/bash
# fe_reduce_cols with two carry chains side by side: arithmetic tests, then the bench (headline, steady state, MSM, prover)
cd "$(dirname "$0")/../../.."
python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu > /tmp/b.json 2>/tmp/b.err
  python3 -c "
import json
d=json.loads(open('/tmp/b.json').readline()); print('rep $rep: value %.0f steady %.0f latency %s msm %.0f prover %.0f %.0f valu %s' % (d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['msm_2p20']['pairs_per_s'], d['prover']['proofs_per_s'], d['prover_1024_constraints']['proofs_per_s'], d['roofline']['step'].get('valu_issue_frac')))
print('   kernels in flight', {k: round(v,3) for k,v in sorted(d['roofline'].get('step',{}).get('kernel_ms_in_flight',{}).items())} )"
done
