#!/bin/bash
cd $GRAFT_REPO_ROOT
export ZKGPU_BENCH_SHARE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --lean > gpurun_out/n2_config2.json 2> gpurun_out/n2_config2.err; python -c "import sys,json; d=json.loads(open('gpurun_out/n2_config2.json').readline()); print(d['value'], d['n_gpus'], d['config']['exchange'], d['config']['steps_per_exchange'])" || tail -5 gpurun_out/n2_config2.err
unset ZKGPU_BENCH_SHARE_GPU
python bench.py --lean --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])"
