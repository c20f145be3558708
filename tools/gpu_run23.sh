#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-steady > gpurun_out/b23.json 2> gpurun_out/b23.err
python -c "
import json; d=json.loads(open('gpurun_out/b23.json').readline()); print(d['value'], d['tx_verify']); print(d['config'].get('lanes'), d['config'].get('ranks'))"
timeout 600 python -m pytest tests/test_gpu_block.py -m gpu -x -q 2>&1 | tail -2
