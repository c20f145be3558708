#!/bin/bash
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
for sl in 0 1 2; do
  ZKGPU_MSM_SLICES=$sl rocprofv3 --kernel-trace --output-format csv -d gpurun_out/msmtl_$sl -- python3 tools/msm_bench.py > /dev/null 2> gpurun_out/msmtl_$sl.err
  f=$(ls gpurun_out/msmtl_$sl/*/*_kernel_trace.csv | head -1)
  echo "== slices=$sl"; python3 tools/msm_timeline.py $f
done
