#!/bin/bash
cd $GRAFT_REPO_ROOT
for T in benched_arrangement mocked_world never_run_over transactions_verified 1024_fixture exchange_step; do
  echo "== $T"; python -X faulthandler -m pytest tests -m gpu -x -q -k "$T" 2>&1 | tail -4
done
for S in 20 200; do
  echo "== merge 10240 steps $S"
  python bench.py --steps $S --warmup 5 --lean 2>gpurun_out/err_$S.txt > gpurun_out/bench_$S.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$S.json").readline()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_ms_in_flight"])
PY
done
for M in 8192 12288; do
  echo "== merge $M steps 200 / 20"
  for S in 200 20; do python bench.py --steps $S --warmup 5 --lean --merge $M 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; done
done
