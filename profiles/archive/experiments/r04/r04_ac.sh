#!/bin/bash
# the single 2^20 multiplication in two slices (run_sliced) against the unsliced chain, same box; then the MSM parity tests
cd "$(dirname "$0")/../../.."
for rep in 1 2 3; do
  for sl in 0 1 2; do
    echo -n "slices=$sl rep=$rep: "
    ZKGPU_MSM_SLICES=$sl python3 tools/msm_bench.py 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d.get('pairs_per_s'), d.get('ms'), d.get('ms_with_kernel_events'), d.get('kernel_ms_sum'))
"
  done
done
python3 -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -3
