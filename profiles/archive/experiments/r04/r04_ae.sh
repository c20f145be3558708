#!/bin/bash
# k_bucket_accumulate with the next row prefetched into LDS (global_load_lds) against the register form, same box
cd "$(dirname "$0")/../../.."
for rep in 1 2 3; do
  for g in 0 1; do
    echo -n "glds=$g rep=$rep: "
    ZKGPU_ACC_GLDS=$g python3 tools/msm_bench.py 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d.get('pairs_per_s'), d.get('ms'), d.get('ms_with_kernel_events'), d['kernel_ms'].get('k_bucket_accumulate'), d['kernel_ms'].get('k_decompress'))
"
  done
done
ZKGPU_ACC_GLDS=1 python3 -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -3
