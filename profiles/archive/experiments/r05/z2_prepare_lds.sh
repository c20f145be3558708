#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O
for KB in 40 48; do
  ZKGPU_AB_PREP_LDS_KB=$KB timeout 600 python3 bench.py --solo --steps 20 > $O/solo_lds$KB.json 2> $O/solo_lds$KB.err
  grep "k_prepare:" $O/solo_lds$KB.err | sort | uniq -c
  python3 -c "
import json;d=json.loads(open('$O/solo_lds$KB.json').read().strip().splitlines()[-1]);print($KB, {k:round(v,4) for k,v in d['solo_kernel_ms'].items()})"
done
