#!/bin/bash
# full GPU test log + where k_prepare's cycles go (instruction cache, LDS, waits), every kernel alone on the chip
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_gpu_tests.log 2>&1; tail -3 gpurun_out/r03b_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_avail.txt 2>&1
grep -c . $R/gpurun_out/counters_avail.txt
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  N=$(echo $P | cut -d" " -f1)
  D=$R/gpurun_out/pmcx_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --solo --steps 6 > /dev/null 2> $D.err
  tail -2 $D.err
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcx_*')):
    if d.endswith('.err'): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0]
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        for k in ('k_prepare', 'k_small_accumulate', 'k_points_tables', 'k_static_accumulate'):
            for kk in acc:
                if kk.startswith(k):
                    print(d.split('/')[-1], kk[:40], {c: '%.3g' % v for c, v in acc[kk].items()})
PY
