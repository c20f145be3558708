#!/bin/bash
# prover: k_static_row_sums with eight lanes per row (eight rows per wavefront) against a wavefront per row
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/row_sums.jsonl
timeout 1800 python3 -m pytest tests/test_gpu_verifier.py tests/test_gpu_block.py -m gpu -x -q -k "prov or Prov" > $O/tests_row_sums.log 2>&1; echo "tests rc=$?" >> $O/tests_row_sums.log
grep -E "passed|failed" $O/tests_row_sums.log
for R in 1 2; do
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child program 1024 16 >> $O/row_sums.jsonl 2>/dev/null
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child cloak 2048 16 >> $O/row_sums.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/row_sums.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child program 4096 16 >> $O/row_sums.jsonl 2>/dev/null
done; done
python3 -c "
import json
for l in open('$O/row_sums.jsonl'):
    d=json.loads(l); print(d['lib'], d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'], {k:v for k,v in d['kernel_ms'].items() if 'row_sums' in k or 'encode' in k})"
