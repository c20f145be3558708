#!/bin/bash
# Run on the GPU box:  bash tools/pmc_breakdown.sh <tag>
# Where a kernel's cycles go, every kernel alone on the chip (bench.py --solo): scalar against vector instructions,
# instruction cache, LDS, waits -- separate PMC passes as gpurun requires.  Then: python tools/pmc_breakdown.py <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  N=$(echo $P | cut -d" " -f1)
  D=$R/gpurun_out/pmcx_${TAG}_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --solo --steps 6 > /dev/null 2> $D.err
done
