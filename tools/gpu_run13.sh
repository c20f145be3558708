#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --lean --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('$*', '->', round(d['value']/1e6,3))"; }
run
run --horner-mode 1
run --horner-mode 2
run --group 24
run --group 32
run --group 32 --horner-mode 1
run --merge 8192
run --merge 12288
run --merge 16384 --tickets 96
run --inflight 4
run --inflight 6
run --inflight 7
run --bad-every 0
run
