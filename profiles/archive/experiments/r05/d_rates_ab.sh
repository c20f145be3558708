#!/bin/bash
# round 5, fourth GPU call: per-opcode VALU issue costs; A/B of the column-serial field product (inline-asm multiply-add chains);
# larger prover calls; a kernel trace of a sliced prover call
O=gpurun_out/r05d; mkdir -p $O
R=$PWD
(cd tools/ubench && for w in 1 2 4; do ./valu_ops $w > $R/$O/valu_ops_w$w.txt 2>&1; done)
for L in tree r05cs; do
  if [ $L = tree ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$R/build/ab/$L/libzkgpu.so; fi
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sweep > $O/bench_$L.json 2> $O/bench_$L.err; echo "bench $L rc=$?" >> $O/rc.txt
  timeout 600 python3 bench.py --lean --steps 200 --warmup 20 > $O/bench200_$L.json 2> $O/bench200_$L.err; echo "bench200 $L rc=$?" >> $O/rc.txt
done
unset ZKGPU_LIB
timeout 1500 python3 tools/prover_sweep.py big > $O/prover_big.jsonl 2> $O/prover_big.err; echo "big rc=$?" >> $O/rc.txt
cd /tmp && export TMPDIR=/tmp
ZKGPU_PROVER_SLICES=4 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_prover -- python3 $R/tools/prover_profile.py 4096 > $R/$O/trace_prover.txt 2> $R/$O/trace_prover.err
cd $R; find $O/trace_prover -name "*kernel_trace.csv" -size +60M -delete
cat $O/rc.txt; cat $O/valu_ops_w4.txt; cut -c1-200 $O/prover_big.jsonl
