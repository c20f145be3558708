#!/bin/bash
# round 5, second GPU call: config 4 at N = 8 on one GPU again (narrow tables), then the prover grid (slices x pipeline x batch x table width)
O=gpurun_out/r05b; mkdir -p $O
ZKGPU_BENCH_SHARE_GPU=1 ZKGPU_BENCH_TRY_RCCL=1 timeout 900 python3 bench.py --config 4 --gpus 8 --steps 6 --warmup 2 > $O/n8_config4.json 2> $O/n8_config4.err; echo "n8c4 rc=$?" >> $O/rc.txt
timeout 2400 python3 tools/prover_sweep.py > $O/prover_sweep.jsonl 2> $O/prover_sweep.err; echo "sweep rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/prover_sweep.jsonl | cut -c1-330
