#!/bin/bash
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/ipa_inline.jsonl
timeout 900 python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_verifier.py -m gpu -x -q -k "arith or prov or Prov" > $O/tests_ipa2.log 2>&1; echo "tests rc=$?" >> $O/tests_ipa2.log
tail -2 $O/tests_ipa2.log
for R in 1 2; do
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child program 1024 16 >> $O/ipa_inline.jsonl 2>/dev/null
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child cloak 2048 16 >> $O/ipa_inline.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/ipa_inline.jsonl 2>/dev/null
done; done
python3 -c "
import json
for l in open('$O/ipa_inline.jsonl'):
    d=json.loads(l); print(d['lib'], d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'], {k:v for k,v in d['kernel_ms'].items() if 'ipa' in k or 'encode' in k})"
