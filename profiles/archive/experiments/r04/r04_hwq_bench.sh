#!/bin/bash
# which GPU_MAX_HW_QUEUES should the library ask for?  (a) the 32768-per-call transaction lottery at 17 / 18 / 19, (b) the
# bench's headline, steady state and transaction leg at 16 / 18 / 24, (c) config 4
cd "$(dirname "$0")/../../.."
for q in 17 18 19; do
  for rep in 1 2 3 4; do
    echo -n "tx q=$q rep=$rep: "
    GPU_MAX_HW_QUEUES=$q python3 tools/tx_call_profile.py 32768 2>&1 | grep "^call\|Error\|error" | sed 's/call \([0-9]\): \([0-9.]*\) ms.*/\2/' | tr '\n' ' '
    echo
  done
done
for rep in 1 2; do
  for q in 16 18 24; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu --no-msm > /tmp/b.json 2>/tmp/b.err
    python3 - $q $rep <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").readline())
t = d.get("tx_verify", {})
print("bench q=%s rep=%s value %.0f steady %.0f host %s lat %s tx8192 %s [%s] tx32768 %s [%s] inflight %s" % (sys.argv[1], sys.argv[2], d["value"],
      d.get("steady_state", {}).get("tx_per_s", 0), d.get("host_memory", {}).get("tickets", {}).get("tx_per_s"), d.get("latency_one_batch_ms"),
      t.get("ms_8192_per_call"), t.get("ms_8192_min_max"), t.get("ms_32768_per_call"), t.get("ms_32768_min_max"), t.get("in_flight", {}).get("tx_per_s")))
PY
  done
done
for q in 16 18 24; do
  for rep in 1 2; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --config 4 > /tmp/b.json 2>/tmp/b.err
    python3 -c "
import json
d = json.loads(open('/tmp/b.json').readline()); print('config4 q=$q rep=$rep value %.0f' % d['value'])"
  done
done
