#!/bin/bash
# k_prepare with the factor u folded into the shared factors of the generator loop: parity, solo time, instruction count
O=gpurun_out/r05z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1800 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_verifier.py -m gpu -x -q > $O/tests4.log 2>&1; echo "tests rc=$?" >> $O/tests4.log
timeout 600 python3 bench.py --solo --steps 20 > $O/solo_ufold.json 2> $O/solo_ufold.err
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/r05z/pmc_ufold
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --solo --steps 6 > /dev/null 2> $D.err
cd $R
tail -2 $O/tests4.log
python3 - <<'PY'
import json,csv,glob,collections
d=json.loads(open('gpurun_out/r05z/solo_ufold.json').read().strip().splitlines()[-1]);print({k:round(v,4) for k,v in d['solo_kernel_ms'].items()})
for f in glob.glob('gpurun_out/r05z/pmc_ufold/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='SQ_INSTS_VALU': acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'prepare' in k or 'small_acc' in k: print(k, len(v), sum(v[-8:])/len(v[-8:]))
PY
