#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr200 -- python3 $R/bench.py --lean --steps 200 --warmup 5 > $R/gpurun_out/tr200.json 2> $R/gpurun_out/tr200.err)
python tools/overlap.py $(ls gpurun_out/tr200/*/*_kernel_trace.csv | head -1)
python -c "import json; d=json.loads(open('gpurun_out/tr200.json').readline()); print('value under rocprof', d['value'])"
echo "== config 4 lanes sweep"
for Q in 24 64; do for N in 6 7 8 9 10; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --config 4 --lean --inflight $N --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hwq $Q lanes $N', d['value'], d['ms_per_step'])"
done; done
echo "== config 2 tickets lanes sweep (steps 200)"
for N in 4 5 6 7 8; do python bench.py --lean --steps 200 --warmup 5 --inflight $N 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lanes $N', d['value'])"; done
echo "== the driver's command, twice"
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full_$i.json 2> gpurun_out/full_$i.err; python - <<PY
import json
d=json.loads(open("gpurun_out/full_$i.json").readline())
print(d["value"], d.get("latency_one_batch_ms"), d.get("steady_state",{}).get("tx_per_s"), d.get("hbm_copy"))
m=d.get("msm_2p20",{}); print("msm", m.get("pairs_per_s"), m.get("ms"), m.get("kernel_ms_sum"), m.get("roofline",{}).get("kernel"), m.get("valu_issue_frac"))
print("tx", d.get("tx_verify")); print("sweep", d["setup"].get("table_bits_sweep"))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
tail -2 gpurun_out/full_$i.err; done
