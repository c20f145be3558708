#!/bin/bash
# round 5, sixth GPU call: opcode costs at 1/2/3/4/8 waves per SIMD with the conditional-move variants; A/B of the column-serial
# field product (aliasing fixed): the arithmetic and MSM tests on the variant, then bench x2 each
O=gpurun_out/r05f; mkdir -p $O
R=$PWD
(cd tools/ubench && for w in 1 2 3 4 8; do ./valu_ops $w > $R/$O/valu_ops_w$w.txt 2>&1; done)
ZKGPU_LIB=$R/build/ab/r05cs/libzkgpu.so timeout 900 python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_msm.py tests/test_gpu_verifier.py -x -q > $O/cs_tests.log 2>&1; echo "cs tests rc=$?" >> $O/rc.txt
for rep in 1 2 3; do
  for L in tree r05cs; do
    if [ $L = tree ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$R/build/ab/$L/libzkgpu.so; fi
    timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep > $O/bench_${L}_$rep.json 2> $O/bench_${L}_$rep.err; echo "bench $L $rep rc=$?" >> $O/rc.txt
  done
done
unset ZKGPU_LIB
cat $O/rc.txt; tail -3 $O/cs_tests.log; cat $O/valu_ops_w3.txt | tail -12
