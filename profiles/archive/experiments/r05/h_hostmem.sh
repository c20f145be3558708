#!/bin/bash
# round 5: host-memory tickets copied to HBM ticket by ticket (twins of the staging areas): the ticket tests, then the bench line x2
O=gpurun_out/r05h; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_block.py -x -q -k "host_memory or newest_first or tickets or blocks" > $O/ticket_tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
for rep in 1 2; do
  /usr/bin/time -v -o $O/bench_time_$rep.txt timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; tail -3 $O/ticket_tests.log; grep -E "Elapsed|Maximum resident" $O/bench_time_1.txt
