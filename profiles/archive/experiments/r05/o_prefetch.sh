#!/bin/bash
# round 5: k_small_accumulate with the next point's row fetched during the current addition (164 registers, three wavefronts per SIMD)
O=gpurun_out/r05o; mkdir -p $O
R=$PWD
ZKGPU_LIB=$R/build/ab/r05pf/libzkgpu.so timeout 900 python3 -m pytest tests/test_gpu_verifier.py tests/test_gpu_msm.py -x -q > $O/pf_tests.log 2>&1; echo "pf tests rc=$?" >> $O/rc.txt
for rep in 1 2 3; do
  for L in tree r05pf; do
    if [ $L = tree ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$R/build/ab/$L/libzkgpu.so; fi
    timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep --no-msm > $O/bench_${L}_$rep.json 2> $O/bench_${L}_$rep.err; echo "bench $L $rep rc=$?" >> $O/rc.txt
  done
done
unset ZKGPU_LIB
cat $O/rc.txt; tail -2 $O/pf_tests.log
