#!/bin/bash
# end of round 5: the GPU tier, smoke, every profile of tools/profile_bench.sh, then the driver's command three times
# (after the profile passes: bench.py reads the counter tables profiles/pmc_*.json, which tools/collect_profiles.py makes HERE
# from those passes -- the committed line is the first of a later call, see profiles/INDEX.md)
O=gpurun_out/r05fin; mkdir -p $O
find gpurun_out -path "*_r05*" -type f -delete 2>/dev/null
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/gpu_tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/profile_bench.sh r05 > $O/profile.log 2>&1
for i in 1 2 3; do timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags_$i.json 2> $O/bench_driverflags_$i.err; done
grep -E "passed|failed" $O/gpu_tests.log; tail -1 $O/smoke.log
python3 -c "
import json
for i in (1,2,3):
    d=json.loads(open('$O/bench_driverflags_%d.json'%i).read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['tx_per_s'], d['latency_one_batch_ms'], d['roofline']['frac'], d['prover']['proofs_per_s'] if 'proofs_per_s' in d['prover'] else d['prover'].get('device',{}).get('proofs_per_s'))"
