#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_block.py -x -q -k "host_memory or newest_first" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
for rep in 1 2 3 4 5 6; do
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep --no-msm > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?" >> $O/rc.txt
done
cat $O/rc.txt
