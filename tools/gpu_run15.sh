#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace15 -- python3 $R/bench.py --lean --steps 200 --warmup 10 > $R/gpurun_out/trace15.json 2> $R/gpurun_out/trace15.err
cd $R
python tools/overlap.py $(ls gpurun_out/trace15/*/*kernel_trace.csv | head -1)
python -c "
import json; d=json.loads(open('gpurun_out/trace15.json').readline()); print(d['value'])"
