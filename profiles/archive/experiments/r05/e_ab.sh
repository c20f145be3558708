#!/bin/bash
# round 5, fifth GPU call: opcode costs (two lengths differenced); A/B of the column-serial field product (fenced sums);
# the wavefront transcript for a batch that finds the device idle; spread of the calls-in-flight leg over fresh processes
O=gpurun_out/r05e; mkdir -p $O
R=$PWD
(cd tools/ubench && for w in 1 2 4; do ./valu_ops $w > $R/$O/valu_ops_w$w.txt 2>&1; done)
ZKGPU_LIB=$R/build/ab/r05cs/libzkgpu.so timeout 600 python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_msm.py -x -q > $O/cs_arith_tests.log 2>&1; echo "cs tests rc=$?" >> $O/rc.txt
for L in tree r05cs; do
  if [ $L = tree ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$R/build/ab/$L/libzkgpu.so; fi
  for rep in 1 2; do
    timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep > $O/bench_${L}_$rep.json 2> $O/bench_${L}_$rep.err; echo "bench $L $rep rc=$?" >> $O/rc.txt
  done
done
unset ZKGPU_LIB
for rep in 1 2; do
  ZKGPU_IDLE_COOP=1 timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep --no-msm > $O/bench_idlecoop_$rep.json 2> $O/bench_idlecoop_$rep.err; echo "idlecoop $rep rc=$?" >> $O/rc.txt
done
for i in 1 2 3 4 5 6 7 8; do timeout 300 python3 tools/tx_inflight.py 1024 8 64 >> $O/tx_inflight_default.txt 2>> $O/tx_inflight.err; done
for i in 1 2 3 4 5 6 7 8; do timeout 300 python3 tools/tx_inflight.py 1024 8 64 4 >> $O/tx_inflight_lanes4.txt 2>> $O/tx_inflight.err; done
cat $O/rc.txt; cat $O/valu_ops_w4.txt; tail -3 $O/cs_arith_tests.log
