#!/bin/bash
# Run on the GPU box:  bash tools/profile_bench.sh <tag>
#   1. kernel trace + stats of the timed bench (batches in flight, the driver's flags)     -> prof_<tag>
#   2. kernel trace + stats of `bench.py --solo` (every kernel alone on the chip, device batches of DISTINCT steps: the
#      averages bench.py's roofline.avg_launch_ms must agree with)                          -> solo_<tag>
#   3. kernel trace + stats of the 2^20 MSM pipeline (tools/msm_bench.py)                   -> msm_<tag>
#   4. PMC passes, separate runs as gpurun requires (no trace domains beside --kernel-trace) -> pmc_<tag>_*
# then:  python tools/collect_profiles.py <tag>   (here, after the call) copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --lean --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/solo_$TAG -- python3 $R/bench.py --solo --steps 20 > $R/gpurun_out/solo_${TAG}_bench.json 2> $R/gpurun_out/solo_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/msm_$TAG -- python3 $R/tools/msm_bench.py > $R/gpurun_out/msm_${TAG}.json 2> $R/gpurun_out/msm_$TAG.err
# 5. the side paths: the device prover (2048 cloak proofs per call), the 1032-constraint program, one call of 8192 serialized
#    transactions -- kernel trace + stats each, and one SQ_INSTS_VALU pass each (counter evidence for them)
for K in "prover tools/prover_profile.py" "proverprog tools/prover_program_profile.py" "tx tools/tx_call_profile.py"; do
  N=$(echo $K | cut -d" " -f1); S=$(echo $K | cut -d" " -f2)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${N}_$TAG -- python3 $R/$S > $R/gpurun_out/${N}_${TAG}.txt 2> $R/gpurun_out/${N}_$TAG.err
  D=$R/gpurun_out/pmc${N}_${TAG}_SQ_WAVE_CYCLES
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $D -- python3 $R/$S > /dev/null 2> $D.err
done
for P in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  N=$(echo $P | cut -d" " -f1)
  D=$R/gpurun_out/pmc_${TAG}_$N
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --solo --steps 6 > /dev/null 2> $D.err
  D=$R/gpurun_out/pmcmsm_${TAG}_$N
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/tools/msm_bench.py > /dev/null 2> $D.err
done
ls $R/gpurun_out | head -60
