#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/bigger_calls.jsonl
for S in 4 6 8; do ZKGPU_PROVER_SLICES=$S timeout 900 python3 tools/prover_sweep.py child cloak 16384 16 >> $O/bigger_calls.jsonl 2>/dev/null; done
for S in 4 8; do ZKGPU_PROVER_SLICES=$S timeout 900 python3 tools/prover_sweep.py child program 8192 16 >> $O/bigger_calls.jsonl 2>/dev/null; done
for S in 4; do ZKGPU_PROVER_SLICES=$S timeout 900 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/bigger_calls.jsonl 2>/dev/null; done
python3 -c "
import json
for l in open('$O/bigger_calls.jsonl'):
    d=json.loads(l); print(d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'])"
