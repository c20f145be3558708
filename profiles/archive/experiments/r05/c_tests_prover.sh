#!/bin/bash
# round 5, third GPU call: the whole GPU test tier on the library without hook exports, the prover with fewer
# k_static_accumulate workgroups per CU, a default bench line
O=gpurun_out/r05c; mkdir -p $O
timeout 2400 python3 tools/prover_sweep.py lds > $O/prover_lds.jsonl 2> $O/prover_lds.err; echo "lds rc=$?" >> $O/rc.txt
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 5 $O/gpu_tests.log; cut -c1-260 $O/prover_lds.jsonl
