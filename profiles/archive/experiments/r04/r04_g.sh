#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_block.py -m gpu -x -q -k "msm" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/msm_r04g -- python3 $R/tools/msm_bench.py > $R/gpurun_out/msm_r04g.json 2> $R/gpurun_out/msm_r04g.err
cat $R/gpurun_out/msm_r04g.json
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/msm_r04g/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-40s %4s %10.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
} > $R/gpurun_out/r04g_msm.txt 2>&1
cat $R/gpurun_out/r04g_msm.txt
