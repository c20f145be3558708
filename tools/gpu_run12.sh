#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "verify or cloak or block or bench or prepare or r1cs" > gpurun_out/t12.log 2>&1; tail -3 gpurun_out/t12.log
python bench.py --solo --steps 10 > gpurun_out/solo12.json 2>gpurun_out/solo12.err; python -c "
import json; d=json.loads(open('gpurun_out/solo12.json').readline()); print({k:v for k,v in d['solo_kernel_ms'].items()})"
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep > gpurun_out/b12_$i.json 2> gpurun_out/b12_$i.err; python -c "
import json; d=json.loads(open('gpurun_out/b12_$i.json').readline()); print('value', d['value'], 'steady', d.get('steady_state',{}).get('tx_per_s'), 'lat', d.get('latency_one_batch_ms'))"; done
