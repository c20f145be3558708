#!/bin/bash
# round 5, first GPU call: the bounded bring-up (stall rehearsals), the livelock test, the benched-layout parity tests,
# and the two N = 8 rehearsal lines on one GPU
O=gpurun_out/r05a; mkdir -p $O
python3 -m pytest tests/test_launch.py -m gpu -x -q > $O/launch_tests.log 2>&1; echo "launch rc=$?" >> $O/rc.txt
python3 -m pytest tests/test_gpu_block.py -x -q -k "benched_arrangement or host_memory or newest_first" > $O/block_tests.log 2>&1; echo "block rc=$?" >> $O/rc.txt
ZKGPU_BENCH_SHARE_GPU=1 ZKGPU_BENCH_TRY_RCCL=1 timeout 1200 python3 bench.py --gpus 8 --steps 20 --warmup 5 > $O/n8_config2.json 2> $O/n8_config2.err; echo "n8c2 rc=$?" >> $O/rc.txt
ZKGPU_BENCH_SHARE_GPU=1 ZKGPU_BENCH_TRY_RCCL=1 timeout 1200 python3 bench.py --config 4 --gpus 8 --steps 6 --warmup 2 > $O/n8_config4.json 2> $O/n8_config4.err; echo "n8c4 rc=$?" >> $O/rc.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "n1 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/launch_tests.log $O/block_tests.log; tail -c 600 $O/n8_config2.err
