#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
for rep in 1 2; do
  S=$(date +%s.%N)
  timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$? seconds=$(echo "$(date +%s.%N) - $S" | bc)" >> $O/rc2.txt
done
cat $O/rc2.txt
