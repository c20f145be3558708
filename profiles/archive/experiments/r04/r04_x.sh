#!/bin/bash
# after the library's default went to 18 hardware queues: the transaction call per process (no variable exported), calls in
# flight with one and two rounds, the bench's full default line, the GPU tests
cd "$(dirname "$0")/../../.."
for rep in 1 2 3 4; do
  for n in 8192 32768; do
    echo -n "tx n=$n rep=$rep: "
    python3 tools/tx_call_profile.py $n 2>&1 | grep "^call\|Error\|error" | sed 's/call \([0-9]\): \([0-9.]*\) ms.*/\2/' | tr '\n' ' '
    echo
  done
done
for r in 1 2; do
  for rep in 1 2; do
    echo "rounds=$r rep=$rep"; ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 1024 8 64 2>&1 | tail -1
    ZKGPU_TX_ROUNDS=$r python3 tools/tx_inflight.py 4096 4 32 2>&1 | tail -1
  done
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04x_bench_driverflags.json 2> gpurun_out/r04x_bench_driverflags.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04x_bench_driverflags.json").readline())
t = d.get("tx_verify", {})
print("bench value %.0f steady %.0f host %s lat %s" % (d["value"], d.get("steady_state", {}).get("tx_per_s", 0), d.get("host_memory", {}).get("tickets", {}).get("tx_per_s"), d.get("latency_one_batch_ms")))
print("tx", {k: v for k, v in t.items() if k.startswith("ms_") or k.startswith("tx_per_s") or k == "in_flight"})
print("prover", d["prover"]["proofs_per_s"], d["prover_1024_constraints"]["proofs_per_s"], "msm", d["msm_2p20"]["pairs_per_s"], "lanes", d["config"].get("lanes"), d["config"].get("hw_queues"))
PY
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04x_gpu_tests.log; cat gpurun_out/r04x_gpu_tests.log
