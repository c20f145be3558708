#!/bin/bash
# the prover's inner-product rounds with one lane per proof (k_pv_ipa_lanes) against one workgroup per proof (the previous build)
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/ipa_lanes.jsonl
timeout 1500 python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_verifier.py tests/test_gpu_block.py -m gpu -x -q -k "arith or prov or Prov" > $O/tests_ipa.log 2>&1; echo "tests rc=$?" >> $O/tests_ipa.log
tail -3 $O/tests_ipa.log
for R in 1 2; do
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/ipa_lanes.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child program 4096 16 >> $O/ipa_lanes.jsonl 2>/dev/null
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child program 1024 16 >> $O/ipa_lanes.jsonl 2>/dev/null
done; done
python3 -c "
import json
for l in open('$O/ipa_lanes.jsonl'):
    d=json.loads(l); print(d['lib'], d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'], d['kernel_ms'])"
