#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --lean "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'])"; }
echo "== random corruption positions: baseline"
run --steps 200 --warmup 5
run --steps 20 --warmup 5
echo "== horner modes (steady)"
for H in 1 2; do run --steps 200 --warmup 5 --horner-mode $H; done
echo "== group sizes (steady)"
for G in 8 12 24 32; do run --steps 200 --warmup 5 --group $G; done
echo "== locate modes"
for M in 1 2 3; do run --steps 200 --warmup 5 --locate-mode $M; done
echo "== no corruption"
run --steps 200 --warmup 5 --bad-every 0
echo "== config 4 blocks in flight 2, lanes"
for N in 6 7 8 9 10; do python bench.py --config 4 --lean --inflight $N --blocks-in-flight 2 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lanes $N blocks 2', d['value'], d['ms_per_step'])"; done
python -m pytest tests -m gpu -x -q -k "benched_arrangement or table_width" 2>&1 | tail -3
