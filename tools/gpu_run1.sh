#!/bin/bash
# first GPU run of round 3: the new tests, then the merge sweep of the headline bench
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "benched_arrangement or mocked_world or never_run_over or transactions_verified or 1024_fixture or exchange_step" 2>&1 | tail -15
for M in 8192 10240 12288; do for S in 20 200; do
  echo "== merge $M steps $S"
  python bench.py --steps $S --warmup 5 --merge $M --lean 2>gpurun_out/err_${M}_$S.txt > gpurun_out/bench_${M}_$S.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${M}_$S.json").readline()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_ms_solo"]); print(d["kernel_ms_in_flight"])
PY
done; done
tail -3 gpurun_out/err_10240_20.txt
