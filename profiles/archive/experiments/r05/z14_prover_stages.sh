#!/bin/bash
# prover: every phase as stages -- one lane per proof where a stage is one thread's work, a workgroup per proof elsewhere
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/stages.jsonl
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/tests_stages.log 2>&1; echo "tests rc=$?" >> $O/tests_stages.log
grep -E "passed|failed" $O/tests_stages.log
for R in 1 2; do
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child program 1024 16 >> $O/stages.jsonl 2>/dev/null
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child cloak 2048 16 >> $O/stages.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/stages.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child program 4096 16 >> $O/stages.jsonl 2>/dev/null
done; done
python3 -c "
import json
for l in open('$O/stages.jsonl'):
    d=json.loads(l); print(d['lib'], d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'], {k:v for k,v in d['kernel_ms'].items() if 'pv_' in k})"
