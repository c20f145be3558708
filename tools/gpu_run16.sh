#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('$*', '->', round(d['value']/1e6,3))"; }
for m in 10240 4096 5120 7168 3072 20480; do run --merge $m; run --merge $m; done
