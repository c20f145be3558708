#!/bin/bash
cd $GRAFT_REPO_ROOT
for Q in 4 6 8 12 16 24; do
  export GPU_MAX_HW_QUEUES=$Q
  A=$(python bench.py --lean --steps 200 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['merged_device_batches']['lanes'])")
  B=$(python bench.py --lean --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])")
  C=$(python bench.py --config 4 --lean --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['calls_in_flight'])")
  D=$(python tools/msm_bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['pairs_per_s'], d['ms'], d['kernel_ms_sum'])")
  echo "hwq $Q | config2 200: $A | 20: $B | config4: $C | msm: $D"
done
