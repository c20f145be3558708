#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 0 45000 54000; do
  export ZKGPU_PREP_LDS=$L
  A=$(python bench.py --lean --steps 200 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], {k:v for k,v in d['kernel_ms_solo'].items() if k in ('k_prepare','k_small_accumulate')})")
  echo "prep lds $L: $A"
done
