#!/bin/bash
# round 5: quad_double / quad_add with fewer carries and two-way selects: the GPU test tier, then the bench three times
O=gpurun_out/r05l; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
for rep in 1 2 3; do
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; grep -E "passed|failed" $O/gpu_tests.log | tail -2
