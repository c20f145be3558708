#!/bin/bash
# Run on the GPU box:  bash tools/profile_bench.sh <tag>
# Kernel-trace + stats of the default bench, then PMC passes (separate runs, as gpurun requires).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --lean > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err
for P in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  N=$(echo $P | cut -d" " -f1)
  D=$R/gpurun_out/pmc_${TAG}_$N
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --lean --steps 4 --warmup 1 --inflight 1 > /dev/null 2> $D.err
done
ls $R/gpurun_out | head -40
