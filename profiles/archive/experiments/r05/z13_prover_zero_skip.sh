#!/bin/bash
# prover: commitment rows of one make side by side, additions no lane needs skipped, a_R = -1 folded to one digit
O=gpurun_out/r05z; mkdir -p $O; rm -f $O/zero_skip.jsonl
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/tests_zero_skip.log 2>&1; echo "tests rc=$?" >> $O/tests_zero_skip.log
tail -2 $O/tests_zero_skip.log
for R in 1 2; do
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child program 1024 16 >> $O/zero_skip.jsonl 2>/dev/null
  ZKGPU_LIB=$L ZKGPU_PROVER_SLICES=1 timeout 600 python3 tools/prover_sweep.py child cloak 2048 16 >> $O/zero_skip.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child cloak 8192 16 >> $O/zero_skip.jsonl 2>/dev/null
  ZKGPU_LIB=$L timeout 600 python3 tools/prover_sweep.py child program 4096 16 >> $O/zero_skip.jsonl 2>/dev/null
done; done
for V in prev tree; do
  L=""; [ $V = prev ] && L=build/ab/prev/libzkgpu.so
  ZKGPU_LIB=$L timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('bench $V', d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'])"
done
python3 -c "
import json
for l in open('$O/zero_skip.jsonl'):
    d=json.loads(l); print(d['lib'], d['kind'], d['batch'], d['slices'], d['ms'], d['proofs_per_s'], {k:v for k,v in d['kernel_ms'].items() if 'static_acc' in k})"
